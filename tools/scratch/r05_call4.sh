#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05d; mkdir -p $O
timeout -k 10 240 python3 tools/bench_split_mid_rows.py > $O/split_mid_rows.txt 2>&1; echo "mid rows rc=$?" > $O/progress.log
cat $O/split_mid_rows.txt
for th in 1024 4096; do
  PAFC_DISPATCH=split_gemm_min_rows=$th timeout -k 10 240 python3 bench.py --workload c2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c2_split$th.json 2> $O/bench_c2_split$th.err; echo "c2 split$th rc=$?" >> $O/progress.log
done
timeout -k 10 200 python3 bench.py --workload c2 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c2_default.json 2> $O/bench_c2_default.err; echo "c2 default rc=$?" >> $O/progress.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05d/bench_c2_*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/bench_c2_progress.log | tail -12
timeout -k 10 300 python3 tools/rtf_sweep.py --chunks 2000,4000,9000 --batches 1,8 --out $O/sweep_corner_default > $O/sweep_corner_default.log 2>&1
PAFC_DISPATCH=split_gemm_min_rows=1024 timeout -k 10 300 python3 tools/rtf_sweep.py --chunks 2000,4000,9000 --batches 1,8 --out $O/sweep_corner_split1024 > $O/sweep_corner_split1024.log 2>&1
PAFC_DISPATCH=split_gemm_min_rows=1024 timeout -k 10 300 python3 tools/rtf_sweep.py --chunks 2000,4000,9000 --batches 1,8 --merge-frames 180000 --out $O/sweep_corner_merged > $O/sweep_corner_merged.log 2>&1
tail -8 $O/sweep_corner_default.log $O/sweep_corner_split1024.log $O/sweep_corner_merged.log
cat $O/progress.log

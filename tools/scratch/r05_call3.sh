#!/bin/bash
# round 5, GPU call 3: A/B of the step (split walk), full default bench, last-step table, sweep corner with thresholds / merge
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05c; mkdir -p $O
for rep in 1 2; do
  PAFC_SPLIT_WALK=hilohi timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_hilohi_$rep.json 2>> $O/bench_ab.err
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_shared_$rep.json 2>> $O/bench_ab.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05c/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e)
PY
echo "ab done" > $O/progress.log
timeout -k 10 600 python3 bench.py > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err; echo "bench rc=$?" >> $O/progress.log
tail -14 $O/bench_c3_n1.err
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/rp2 -o run --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-extra --no-cpu-baseline > $O/rp2.json 2> $O/rp2.err
python3 tools/prof_last_step.py $O/rp2/run_kernel_trace.csv 45 > $O/bench_last_step_kernels.txt; rm -rf $O/rp2
echo "last step done" >> $O/progress.log
head -40 $O/bench_last_step_kernels.txt
# the launch-bound corner of the sweep: literal batches at the default threshold and with split operands from 2048 rows, then merged
timeout -k 10 300 python3 tools/rtf_sweep.py --chunks 2000,4000,9000 --batches 1,8 --out $O/sweep_corner_default > $O/sweep_corner_default.log 2>&1
PAFC_DISPATCH=split_gemm_min_rows=2048 timeout -k 10 300 python3 tools/rtf_sweep.py --chunks 2000,4000,9000 --batches 1,8 --out $O/sweep_corner_split2048 > $O/sweep_corner_split2048.log 2>&1
PAFC_DISPATCH=split_gemm_min_rows=2048 timeout -k 10 300 python3 tools/rtf_sweep.py --chunks 2000,4000,9000 --batches 1,8 --merge-frames 180000 --out $O/sweep_corner_merged > $O/sweep_corner_merged.log 2>&1
tail -8 $O/sweep_corner_default.log $O/sweep_corner_split2048.log $O/sweep_corner_merged.log
cat $O/progress.log

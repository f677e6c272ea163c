#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05p; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_fused_gpu.py tests/test_encoder_gpu.py -m gpu -q -x -k "split or bf16slot or gemm" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/progress.log
tail -3 $O/pytest.log
for rep in 1 2 3; do
  PAFC_PH_ROW_PLAN=0 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_one_launch_$rep.json 2>> $O/bench.err
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_row_plan_$rep.json 2>> $O/bench.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05p/bench_*.json')):
    try:
        d=json.load(open(f)); k=d['mfma']['kernels']
        w1=[v for n,v in k.items() if '512x2048' in n][0]
        print(f, d['ms_per_step'], d['value'], 'w_1 us', w1['avg_us'])
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/progress.log

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_wkv6_gpu.py tests/test_streaming_gpu.py tests/test_train_step.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/progress.log
tail -3 $O/pytest.log
for rep in 1 2 3; do
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_$rep.json 2>> $O/bench.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05n/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
bash tools/prof_wkv_traffic.sh r05n > $O/wkv_traffic.log 2>&1; echo "wkv traffic rc=$?" >> $O/progress.log
grep -A4 "kernel_avg_duration" $O/wkv_traffic.log | head -8; grep "hbm_bytes_per_launch_corrected\|op_us" $O/wkv_traffic.log
cat $O/progress.log

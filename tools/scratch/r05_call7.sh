#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05g; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_fused_gpu.py tests/test_train_step.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/progress.log
tail -4 $O/pytest.log
for rep in 1 2 3; do
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_$rep.json 2>> $O/bench.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05g/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['roofline']['traffic'])
    except Exception as e: print(f, 'ERR', e)
PY
for st in 3 6 8; do
  timeout -k 10 200 python3 tools/rtf_sweep.py --chunks 2000,4000,9000 --batches 1,4 --streams $st --no-eager-check --out $O/sweep_streams$st > $O/sweep_streams$st.log 2>&1
  echo "== streams $st"; grep chunk $O/sweep_streams$st.log
done
for rep in 1 2; do
  timeout -k 10 300 python3 tools/bench_train_step.py --amp bf16 >> $O/train_own_gemms.jsonl 2>> $O/train.err
done
cut -c1-250 $O/train_own_gemms.jsonl
bash tools/prof_train.sh r05g_train_step_amp > $O/prof_train.log 2>&1; echo "prof train rc=$?" >> $O/progress.log
head -3 $O/prof_train.log
grep -c "Cijk" gpurun_out/prof_train/r05g_train_step_amp_kernels.txt
cat $O/progress.log

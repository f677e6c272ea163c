#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05f; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/progress.log
tail -6 $O/pytest.log
for rep in 1 2; do
  PAFC_TRAIN_OWN_GEMMS=0 timeout -k 10 300 python3 tools/bench_train_step.py --amp bf16 >> $O/train_library_gemms.jsonl 2>> $O/train.err
  timeout -k 10 300 python3 tools/bench_train_step.py --amp bf16 >> $O/train_own_gemms.jsonl 2>> $O/train.err
done
echo "--- library"; cut -c1-260 $O/train_library_gemms.jsonl; echo "--- own"; cut -c1-260 $O/train_own_gemms.jsonl
echo "train ab done" >> $O/progress.log
bash tools/prof_train.sh r05f_train_step_amp > $O/prof_train.log 2>&1; echo "prof train rc=$?" >> $O/progress.log
head -3 $O/prof_train.log
timeout -k 10 600 python3 bench.py > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err; echo "bench rc=$?" >> $O/progress.log
tail -3 $O/bench_c3_n1.err
bash tools/prof_step_hbm.sh r05f_bf16slot > $O/step_hbm_bf16slot.log 2>&1; echo "hbm rc=$?" >> $O/progress.log
bash tools/prof_wkv_traffic.sh r05f > $O/wkv_traffic.log 2>&1; echo "wkv traffic rc=$?" >> $O/progress.log
tail -12 $O/wkv_traffic.log
BENCH_ARGS="--dtype bf16" bash tools/prof_step_hbm.sh r05f_bf16_rb0 > $O/step_hbm_bf16_rb0.log 2>&1; echo "hbm bf16 rc=$?" >> $O/progress.log
PAFC_FFN_ROW_BLOCK=22528 BENCH_ARGS="--dtype bf16" bash tools/prof_step_hbm.sh r05f_bf16_rb22528 > $O/step_hbm_bf16_rb22528.log 2>&1; echo "hbm bf16 rb rc=$?" >> $O/progress.log
cat $O/progress.log

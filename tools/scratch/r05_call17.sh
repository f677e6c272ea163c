#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05q; mkdir -p $O
for m in "uni 12" "uni 18" "bi 18" "bi 24" "bi 30"; do
  set -- $m
  timeout -k 10 420 python3 tools/rtf_sweep.py --direction $1 --num-blocks $2 --out $O/rtf_sweep_bf16slot_rwkv_$1_$2L > $O/rtf_sweep_$1_$2L.log 2>&1; echo "sweep $1 $2 rc=$?" >> $O/progress.log
  grep "chunk   2000 x batch  8\|chunk  60000 x batch  8" $O/rtf_sweep_$1_$2L.log
  head -1 $O/rtf_sweep_bf16slot_rwkv_$1_$2L.jsonl | cut -c1-160
done
cat $O/progress.log

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05j; mkdir -p $O
for shp in "44998 2048 512" "44998 512 2048" "44998 512 512" "44998 1024 512"; do
  timeout -k 10 120 tools/micro/bin/gemm_ph_check il $shp >> $O/interleaved_planes.log 2>&1
done
cat $O/interleaved_planes.log
timeout -k 10 400 python -m pytest tests/test_encoder_gpu.py tests/test_train_step.py -m gpu -q -x -k "graph or transposed or c4 or gradients" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/progress.log
tail -5 $O/pytest.log
timeout -k 10 300 python3 tools/rtf_sweep.py --dtype bf16 --chunks 9000,15000 --batches 8,12,14 --out $O/sweep_bf16_long_batches > $O/sweep_bf16_long_batches.log 2>&1
grep chunk $O/sweep_bf16_long_batches.log
timeout -k 10 300 python3 tools/rtf_sweep.py --chunks 9000,15000,20000 --batches 12,14 --out $O/sweep_bf16slot_long_batches > $O/sweep_bf16slot_long_batches.log 2>&1
grep chunk $O/sweep_bf16slot_long_batches.log
cat $O/progress.log

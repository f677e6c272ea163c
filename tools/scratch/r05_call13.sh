#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_encoder_gpu.py -m gpu -q -x -s -k "five_minute or token_lists or unmasked" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/progress.log
grep "bf16slot 5-minute\|token lists\|passed\|failed\|Error" $O/pytest.log | head -20
cat $O/progress.log

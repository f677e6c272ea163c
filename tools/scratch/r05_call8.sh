#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05h; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_encoder_gpu.py -m gpu -q -x -k "graph or windows or threads" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/progress.log
tail -4 $O/pytest.log
for st in 3 4 6; do
  timeout -k 10 200 python3 tools/rtf_sweep.py --chunks 2000,4000,9000 --batches 1,4,8 --streams $st --out $O/sweep_threads_streams$st > $O/sweep_threads_streams$st.log 2>&1
  echo "== threads, streams $st"; grep chunk $O/sweep_threads_streams$st.log
done
timeout -k 10 200 python3 tools/rtf_sweep.py --chunks 2000,4000,9000 --batches 1,4,8 --streams 3 --no-host-threads --out $O/sweep_one_thread > $O/sweep_one_thread.log 2>&1
echo "== one thread, streams 3"; grep chunk $O/sweep_one_thread.log
cat $O/progress.log

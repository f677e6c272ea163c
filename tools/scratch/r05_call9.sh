#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05i; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/progress.log
tail -4 $O/pytest.log
timeout -k 10 600 python3 bench.py > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err; echo "bench rc=$?" >> $O/progress.log
tail -3 $O/bench_c3_n1.err
PAFC_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --dist-backend gloo --steps 5 --warmup 2 > $O/bench_two_ranks_gloo_one_gpu.json 2> $O/bench_two_ranks.err; echo "2 ranks rc=$?" >> $O/progress.log
cut -c1-400 $O/bench_two_ranks_gloo_one_gpu.json
timeout -k 10 500 python3 tools/rtf_sweep.py --dtype bf16 --out $O/rtf_sweep_bf16 > $O/rtf_sweep_bf16.log 2>&1; echo "sweep bf16 rc=$?" >> $O/progress.log
timeout -k 10 500 python3 tools/rtf_sweep.py --dtype bf16 --merge-frames 180000 --out $O/rtf_sweep_bf16_merged > $O/rtf_sweep_bf16_merged.log 2>&1; echo "sweep bf16 merged rc=$?" >> $O/progress.log
grep "chunk   2000\|chunk   9000" $O/rtf_sweep_bf16.log | head -12
grep "chunk   2000" $O/rtf_sweep_bf16_merged.log | head -6
for s in 1 8 64; do timeout -k 10 200 python3 tools/bench_streaming.py 64 1800 1 $s >> $O/streaming_carry.jsonl 2>> $O/streaming.err; done
cut -c1-200 $O/streaming_carry.jsonl
cat $O/progress.log

#!/bin/bash
# Secondary records of the round on the GPU box (gpurun_out/final2/): c2, the paper's windowed sweep, streaming, training step.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/final2; mkdir -p $O
timeout -k 10 400 python3 bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
echo c2 > $O/progress.log
for cs in 2000 9000 40000; do
  timeout -k 10 300 python3 bench.py --chunk-size $cs --batch-size 8 --steps 5 --warmup 2 --no-extra --no-cpu-baseline >> $O/windowed_sweep.jsonl 2>> $O/windowed.err
done
echo windows >> $O/progress.log
for s in 1 2 8 64; do timeout -k 10 200 python3 tools/bench_streaming.py 64 1800 1 $s >> $O/streaming_carry.jsonl 2>> $O/streaming.err; done
echo streaming >> $O/progress.log
timeout -k 10 400 python3 tools/bench_train_step.py --amp bf16 > $O/train_step.jsonl 2> $O/train_step.err || echo "train step failed" >> $O/progress.log
echo train >> $O/progress.log
ls $O

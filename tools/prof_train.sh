# rocprofv3 kernel trace of the c4 training step (GPU box), one steady-state step summarised per kernel: tools/prof_train.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
O=gpurun_out/prof_train; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace -d $O/t -o $1 --output-format csv -- python3 tools/bench_train_step.py --amp bf16 --steps 3 --warmup 2 > $O/$1.jsonl 2> $O/$1.err
python3 tools/prof_train_last_step.py $(ls $O/t/*$1*kernel_trace.csv | head -1) 70 > $O/$1_kernels.txt
rm -rf $O/t
cat $O/$1.jsonl | tail -1 | cut -c1-300
head -75 $O/$1_kernels.txt

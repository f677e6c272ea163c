#!/bin/bash
# The round's profile set on the GPU box (writes gpurun_out/final/; copy what is judged into profiles/):
#   1. python bench.py (the driver's default command) -> bench_c3_n1.json
#   2. the same command under rocprofv3 --kernel-trace --stats -> kernel_stats_whole_run.csv, bench_c3_n1_under_rocprof.json
#   3. last-step kernel table of a --no-extra run
#   4. MfmaUtil counter pass (--pmc only, no trace domains beside it)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/final; mkdir -p $O
timeout -k 10 500 python3 bench.py > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err
echo "bench done" > $O/progress.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/rp1 -o run --output-format csv -- python3 bench.py > $O/bench_c3_n1_under_rocprof.json 2> $O/rp1.err
cp $O/rp1/run_kernel_stats.csv $O/kernel_stats_whole_run.csv; rm -rf $O/rp1
echo "stats done" >> $O/progress.log
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/rp2 -o run --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-extra --no-cpu-baseline > $O/rp2.json 2> $O/rp2.err
python3 tools/prof_last_step.py $O/rp2/run_kernel_trace.csv 40 > $O/bench_last_step_kernels.txt; rm -rf $O/rp2
echo "last step done" >> $O/progress.log
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/rp3 -- python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $O/rp3.json 2> $O/rp3.err
f=$(ls $O/rp3/*/*counter_collection.csv | head -1)
python3 tools/summarize_mfma_pmc.py $f 25 > $O/bench_c3_mfma_util_pmc.txt; rm -rf $O/rp3
echo "pmc done" >> $O/progress.log
ls $O

#!/bin/bash
# round 5, GPU call 1: GPU tests, default bench, last-step kernel table + MfmaUtil of the headline precision
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05a; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/progress.log
tail -5 $O/pytest.log
timeout -k 10 500 python3 bench.py > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err; echo "bench rc=$?" >> $O/progress.log
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/rp2 -o run --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-extra --no-cpu-baseline > $O/rp2.json 2> $O/rp2.err
python3 tools/prof_last_step.py $O/rp2/run_kernel_trace.csv 40 > $O/bench_last_step_kernels.txt; rm -rf $O/rp2
echo "last step done" >> $O/progress.log
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/rp3 -- python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $O/rp3.json 2> $O/rp3.err
f=$(ls $O/rp3/*/*counter_collection.csv | head -1)
python3 tools/summarize_mfma_pmc.py $f 25 > $O/bench_c3_mfma_util_pmc.txt; rm -rf $O/rp3
echo "pmc done" >> $O/progress.log
cat $O/progress.log; head -c 1500 $O/bench_c3_n1.json

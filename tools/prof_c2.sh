# rocprofv3 kernel statistics of one c2 pass (GPU box): tools/prof_c2.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; export PYTHONPATH=$R
mkdir -p gpurun_out/prof_c2
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c2 -o $1 --output-format csv -- python3 bench.py --workload c2 --steps 1 --warmup 1 --no-cpu-baseline --streams 1 > gpurun_out/prof_c2_$1.json 2> gpurun_out/prof_c2_$1.err
rm -f gpurun_out/prof_c2/$1_kernel_trace.csv
ls gpurun_out/prof_c2

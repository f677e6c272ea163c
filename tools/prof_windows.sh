# rocprofv3 kernel trace of the paper's window sweep shape (2 000-frame windows x batch 8, bench.py --chunk-size 2000) on the
# GPU box, summarised per kernel over the last step: tools/prof_windows.sh <tag> [chunk] [batch]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; export PYTHONPATH=$R
C=${2:-2000}; B=${3:-8}
mkdir -p gpurun_out/prof_win
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/prof_win -o $1 --output-format csv -- python3 bench.py --chunk-size $C --batch-size $B --steps 3 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/prof_win_$1.json 2> gpurun_out/prof_win_$1.err
NB=$(python3 -c "import math; print(math.ceil(179998 / ($C * $B)))")
python3 tools/prof_last_step.py $(ls gpurun_out/prof_win/*$1*kernel_trace.csv | head -1) 40 $NB > gpurun_out/prof_win_$1_last_step.txt
rm -f gpurun_out/prof_win/*kernel_trace.csv
head -50 gpurun_out/prof_win_$1_last_step.txt

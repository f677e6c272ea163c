# HBM bytes per kernel of the last bench step (GPU box): tools/prof_step_hbm.sh <tag>
# three runs of the same command: kernel trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE (counters in passes of their own, no trace
# domains beside them), each under its own timeout
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
O=gpurun_out/step_hbm_$1; rm -rf $O; mkdir -p $O
CMD="python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline ${BENCH_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace -o r --output-format csv -- $CMD > $O/trace.json 2> $O/trace.err
echo trace >> $O/progress.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o r --output-format csv -- $CMD > $O/fetch.json 2> $O/fetch.err
echo fetch >> $O/progress.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/write -o r --output-format csv -- $CMD > $O/write.json 2> $O/write.err
echo write >> $O/progress.log
python3 tools/summarize_step_hbm.py $O $O/summary.txt > /dev/null
rm -rf $O/trace $O/fetch $O/write
cat $O/summary.txt

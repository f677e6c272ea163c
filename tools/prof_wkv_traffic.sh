# HBM traffic + kernel durations of the bidirectional scan at the 30-minute shape (GPU box), summarised into profiles/<tag>_*:
#   tools/prof_wkv_traffic.sh <tag>      four separate rocprofv3 runs of the same command (kernel stats, FETCH_SIZE, WRITE_SIZE, SQ)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
O=gpurun_out/prof_wkv_$1; rm -rf $O; mkdir -p $O
CMD="python3 tools/bench_wkv6_one.py 1 44998 bf16"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- $CMD > $O/stats.log 2>&1
echo stats >> $O/progress.log
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run --output-format csv -- $CMD > $O/fetch.log 2>&1
echo fetch >> $O/progress.log
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o run --output-format csv -- $CMD > $O/write.log 2>&1
echo write >> $O/progress.log
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/sq -o run --output-format csv -- $CMD > $O/sq.log 2>&1
echo sq >> $O/progress.log
# (rocprofv3 -o run writes the csv files straight into each -d directory)
ls $O/stats $O/fetch | head
mkdir -p gpurun_out/wkv_traffic
python3 tools/summarize_wkv_pmc.py $O gpurun_out/wkv_traffic/$1_wkv6_bidir_T44998_bf16 | tail -40

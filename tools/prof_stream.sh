set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_stream
cd $R; export PYTHONPATH=$R
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stream -o s1 --output-format csv -- python3 tools/bench_streaming.py 64 120 1 ${STREAMS:-1} > gpurun_out/r3v_stream_prof.json 2> gpurun_out/r3v_stream_prof.err
ls gpurun_out/prof_stream

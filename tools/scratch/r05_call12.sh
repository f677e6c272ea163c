#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05l; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/progress.log
tail -4 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/progress.log; tail -1 $O/smoke.log
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err; echo "bench rc=$?" >> $O/progress.log
tail -3 $O/bench_c3_n1.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05l/bench_c3_n1.json'))
print(d['value'], d['ms_per_step'], d['dtype'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['traffic'])
e=d['extra']; print('bf16', e['whole_model_bf16']['ms_per_step'], 'c2', e['c2']['audio_sec_per_sec'], 'win', e['windows_2000x8']['audio_sec_per_sec'], e['windows_2000x8']['merged_launches']['audio_sec_per_sec'])
PY
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/rp1 -o run --output-format csv -- python3 bench.py --no-extra --no-cpu-baseline > $O/bench_c3_n1_under_rocprof.json 2> $O/rp1.err
cp $O/rp1/run_kernel_stats.csv $O/kernel_stats_whole_run.csv; rm -rf $O/rp1
head -12 $O/kernel_stats_whole_run.csv | cut -c1-160
cat $O/progress.log

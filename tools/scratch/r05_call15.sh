#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05o; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/progress.log
tail -4 $O/pytest.log
timeout -k 10 400 python3 bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?" >> $O/progress.log
for cs in 2000 9000 40000; do
  timeout -k 10 300 python3 bench.py --chunk-size $cs --batch-size 8 --steps 5 --warmup 2 --no-extra --no-cpu-baseline >> $O/windowed_sweep.jsonl 2>> $O/windowed.err
done
echo windows >> $O/progress.log
for rep in 1 2; do timeout -k 10 300 python3 tools/bench_train_step.py --amp bf16 >> $O/train_step.jsonl 2>> $O/train_step.err; done
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05o/bench_c2.json')); print('c2', d['value'], d['ms_per_step'], d['dtype'])
for l in open('gpurun_out/r05o/windowed_sweep.jsonl'):
    j=json.loads(l); print('win', j['config']['workload'][60:120], j['value'], j['ms_per_step'])
for l in open('gpurun_out/r05o/train_step.jsonl'):
    j=json.loads(l); print('train', j['ms_per_step'], j['value'])
PY
cat $O/progress.log

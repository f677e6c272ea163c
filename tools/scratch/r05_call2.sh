#!/bin/bash
# round 5, GPU call 2: shared-fragment split GEMMs (check + race), scan grid orders, A/B of the step, tests, full bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05b; mkdir -p $O
timeout -k 10 400 tools/micro/bin/gemm_ph_check all > $O/gemm_ph_check.log 2>&1; echo "gemm check rc=$?" > $O/progress.log
tail -22 $O/gemm_ph_check.log
for rep in 1 2; do
  for v in "0 0" "1 0" "0 1" "1 1"; do
    set -- $v
    echo "REVC=$1 ORDER=$2: $(PAFC_WKV6_REVC=$1 PAFC_WKV6_ORDER=$2 timeout -k 10 120 python3 tools/bench_wkv6_one.py 1 44998 bf16 2>&1 | tail -1)" >> $O/scan_orders.log
  done
done
cat $O/scan_orders.log
for rep in 1 2; do
  PAFC_SPLIT_WALK=hilohi timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_hilohi_$rep.json 2>> $O/bench_ab.err
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/bench_shared_$rep.json 2>> $O/bench_ab.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05b/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e)
PY
echo "ab done" >> $O/progress.log
timeout -k 10 700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/progress.log
tail -8 $O/pytest.log
timeout -k 10 500 python3 bench.py > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err; echo "bench rc=$?" >> $O/progress.log
tail -12 $O/bench_c3_n1.err
cat $O/progress.log

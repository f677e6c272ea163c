#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export PYTHONPATH=$R
O=gpurun_out/r05e; mkdir -p $O
# 1. where does the fp32-library c2 pass stop? (threshold back at 16384, stacks dumped after 60 s)
PAFC_DISPATCH=split_gemm_min_rows=16384 PAFC_BENCH_WATCHDOG=60 timeout -k 10 120 python3 bench.py --workload c2 --steps 1 --warmup 1 --no-cpu-baseline > $O/c2_lib_watchdog.json 2> $O/c2_lib_watchdog.err; echo "watchdog rc=$?" > $O/progress.log
tail -40 $O/c2_lib_watchdog.err
# 2. tests
timeout -k 10 700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/progress.log
tail -6 $O/pytest.log
# 3. the default bench line
timeout -k 10 600 python3 bench.py > $O/bench_c3_n1.json 2> $O/bench_c3_n1.err; echo "bench rc=$?" >> $O/progress.log
tail -12 $O/bench_c3_n1.err
# 4. last-step kernel table and MfmaUtil of the headline precision
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/rp2 -o run --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-extra --no-cpu-baseline > $O/rp2.json 2> $O/rp2.err
python3 tools/prof_last_step.py $O/rp2/run_kernel_trace.csv 45 > $O/bench_last_step_kernels.txt; rm -rf $O/rp2
echo "last step done" >> $O/progress.log
head -36 $O/bench_last_step_kernels.txt
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/rp3 -- python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $O/rp3.json 2> $O/rp3.err
f=$(ls $O/rp3/*/*counter_collection.csv | head -1)
python3 tools/summarize_mfma_pmc.py $f 25 > $O/bench_c3_mfma_util_pmc.txt; rm -rf $O/rp3
echo "pmc done" >> $O/progress.log
# 5. the sweep grid: literal schedule, then merged launches
timeout -k 10 500 python3 tools/rtf_sweep.py --out $O/rtf_sweep_bf16slot > $O/rtf_sweep_bf16slot.log 2>&1; echo "sweep rc=$?" >> $O/progress.log
timeout -k 10 500 python3 tools/rtf_sweep.py --merge-frames 180000 --out $O/rtf_sweep_bf16slot_merged > $O/rtf_sweep_bf16slot_merged.log 2>&1; echo "sweep merged rc=$?" >> $O/progress.log
tail -4 $O/rtf_sweep_bf16slot.log; tail -4 $O/rtf_sweep_bf16slot_merged.log
# 6. FFN row blocks (whole-model bf16, the folded-LayerNorm schedule)
for rep in 1 2; do
  for rb in 0 16384 22528; do
    PAFC_FFN_ROW_BLOCK=$rb timeout -k 10 200 python3 bench.py --dtype bf16 --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $O/ffn_rb${rb}_$rep.json 2>> $O/ffn_rb.err
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05e/ffn_rb*.json')):
    try:
        d=json.load(open(f)); print(f, d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/progress.log

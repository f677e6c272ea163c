#!/bin/bash
# Hardware counters of the phase-pipelined GEMM at the 30-minute shapes (GPU box; writes gpurun_out/gemm_pmc/<case>/<pass>.csv):
#   tools/prof_gemm_pmc.sh            -- separate rocprofv3 --pmc passes (no trace domains beside them), 8 launches each
# Round 3's TA_* pass asked for four TA counters at once: rocprofiler refused the configuration ("error code 38: Request exceeds
# the capabilities of the hardware to collect", profiles/r03_gemm_pmc_ta_pass_abort.log) and aborted in its own initialisation,
# 80 ms after start-up and before the harness had launched a kernel -- the profiler's counter validation, not a kernel fault.
# The TA counters are therefore collected ONE per pass (ta1 .. ta4); every pass runs under its own timeout and its log is kept.
# Summarise with tools/summarize_gemm_pmc.py gpurun_out/gemm_pmc profiles/r03x_gemm_ph_pmc.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
BIN=$R/tools/micro/bin/gemm_ph_check
OUT=$R/gpurun_out/gemm_pmc
mkdir -p $OUT
declare -A P
P[sq_wait]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P[sq_inst]="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
P[sq_vmem]="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES"
P[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
P[tcc2]="TCC_READ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum"
P[tcp]="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
P[ta1]="TA_BUSY_sum"
P[ta2]="TA_ADDR_STALLED_BY_TC_CYCLES_sum"
P[ta3]="TA_DATA_STALLED_BY_TC_CYCLES_sum"
P[ta4]="TA_BUFFER_WAVEFRONTS_sum"
for c in ${CASES:-0 1 5}; do
  for p in ${PASSES:-sq_wait sq_inst sq_vmem tcc tcc2 tcp ta1 ta2 ta3 ta4}; do
    timeout -k 5 150 rocprofv3 --pmc ${P[$p]} --output-format csv -d $OUT/c$c/$p -- $BIN pmc $c > $OUT/c${c}_$p.log 2>&1 || echo "pass $p of case $c failed" >> $OUT/failed.txt
    f=$(ls $OUT/c$c/$p/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && cp $f $OUT/c${c}_$p.csv
    rm -rf $OUT/c$c/$p
  done
done
ls $OUT

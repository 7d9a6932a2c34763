#!/bin/bash
# Run on the GPU box: PMC passes over a command, one counter group per pass (never mixed with tracing
# domains other than kernel-trace).  usage: tools/gpu_pmc.sh OUTDIR python3 script.py args...
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -- "$@" > "$R/$OUT/trace.log" 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$R/$OUT/pmc_fetch" -- "$@" > "$R/$OUT/pmc_fetch.log" 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$R/$OUT/pmc_write" -- "$@" > "$R/$OUT/pmc_write.log" 2>&1
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$R/$OUT/pmc_sq" -- "$@" > "$R/$OUT/pmc_sq.log" 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$R/$OUT/pmc_sq2" -- "$@" > "$R/$OUT/pmc_sq2.log" 2>&1
timeout 240 rocprofv3 --pmc TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum --output-format csv -d "$R/$OUT/pmc_ta" -- "$@" > "$R/$OUT/pmc_ta.log" 2>&1
cd "$R" && python3 tools/pmc_summary.py "$OUT"/trace "$OUT"/pmc_* > "$OUT/summary.txt" 2>&1
# keep only the summary and the stats (raw per-dispatch CSVs are large)
# (GRBM_GUI_ACTIVE rides in the SAME pass as SQ_VALU_MFMA_BUSY_CYCLES: mfma_util divides two counters of one run, and the
#  summary takes each pass's kernel durations from that pass's own dispatch timestamps -- VERDICT r3 weak #8)
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete

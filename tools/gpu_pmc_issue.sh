#!/bin/bash
# Run on the GPU box: two extra PMC passes about instruction issue (who holds the wave: VALU / LDS / scalar / misc, LDS FIFOs,
# instruction fetch).  usage: tools/gpu_pmc_issue.sh OUTDIR python3 /abs/script.py args...
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS --output-format csv -d "$R/$OUT/pmc_issue1" -- "$@" > "$R/$OUT/pmc_issue1.log" 2>&1
timeout 240 rocprofv3 --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d "$R/$OUT/pmc_issue2" -- "$@" > "$R/$OUT/pmc_issue2.log" 2>&1
cd "$R" && python3 tools/pmc_summary.py "$OUT"/pmc_issue* > "$OUT/summary_issue.txt" 2>&1
find "$OUT" -name "*counter_collection.csv" -delete

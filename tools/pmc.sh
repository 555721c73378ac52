#!/bin/bash
# Collect PMC counters for one kbench invocation, one counter group per pass (own runs: never
# combined with tracing).  usage: tools/pmc.sh <out-subdir-under-gpurun_out> <kbench args…>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for G in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $G --output-format csv -d "$OUT/p$i" -- python3 "$R/tools/kbench.py" "$@" > "$OUT/p$i.log" 2>&1
done
python3 "$R/tools/pmc_summary.py" "$OUT" | tee "$OUT/summary.txt"

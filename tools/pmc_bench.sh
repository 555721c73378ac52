#!/bin/bash
# PMC counters of the bench command itself (same program, same kernels, same inputs as the line bench.py prints):
# one counter group per rocprofv3 pass, passes never combined with tracing, the program directly after `--`.
#   usage (on the GPU box):  bash tools/pmc_bench.sh <tag> [bench.py args…]      → gpurun_out/<tag>/{summary.txt,traffic.json}
# Copy summary.txt / traffic.json into profiles/ to have them judged (bench.py reads profiles/traffic.json).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# PMC_DRR=1: keep the projector legs in the run (their kernel gets its own traffic.json entry, with SQ_INSTS_VALU for roofline_drr)
ARGS="--no-cpu-baseline $([ "${PMC_DRR:-0}" = 1 ] || echo --no-drr) --ramp-seconds 0 --steps 3 --warmup 1 $*"
i=0
for G in \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $G --output-format csv -d "$OUT/p$i" -- python3 "$R/bench.py" $ARGS > "$OUT/p$i.log" 2>&1
  echo "pass $i ($G): exit $?" >> "$OUT/passes.log"
done
python3 "$R/tools/traffic_from_pmc.py" "$OUT" --bench-args "$ARGS" | tee "$OUT/summary.txt"
# the raw per-dispatch CSVs are large: keep the summaries only
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete

#!/bin/bash
# PMC counters of the C3 training step's kernels (tools/train_bench.py), one counter group per rocprofv3 pass (never with tracing).
#   usage (on the GPU box):  bash tools/pmc_train.sh <tag> [train_bench args…]   → gpurun_out/<tag>/pmc_train_summary.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-kernel-table --ramp-seconds 0 --steps 2 --warmup 1 $*"
i=0
for G in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G --output-format csv -d "$OUT/t$i" -- python3 "$R/tools/train_bench.py" $ARGS > "$OUT/t$i.log" 2>&1
  echo "pass $i ($G): exit $?" >> "$OUT/passes_train.log"
done
python3 - "$OUT" <<'P' | tee "$OUT/pmc_train_summary.txt"
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/t*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0))[:10]:
    print("==", k)
    for c in sorted(acc[k]):
        print(f"   {c:28s} {acc[k][c] / max(1, cnt[k][c]):16.1f}   (per dispatch, {cnt[k][c]} dispatches)")
    a = {c: acc[k][c] / max(1, cnt[k][c]) for c in acc[k]}
    if a.get("SQ_INSTS_MFMA"):
        print(f"   VALU-non-MFMA per MFMA {(a.get('SQ_INSTS_VALU', 0) - a['SQ_INSTS_MFMA']) / a['SQ_INSTS_MFMA']:.2f}   LDS per MFMA {a.get('SQ_INSTS_LDS', 0) / a['SQ_INSTS_MFMA']:.2f}   bank-conflict cycles per LDS inst {a.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, a.get('SQ_INSTS_LDS', 1)):.2f}")
    if a.get("SQ_BUSY_CYCLES") and a.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        print(f"   matrix pipe busy / SQ busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / a['SQ_BUSY_CYCLES']:.3f}")
P
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete

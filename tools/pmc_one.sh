#!/bin/bash
# PMC counters of ONE python command (development aid): bash tools/pmc_one.sh <tag> <python script + args…>
# passes: FETCH_SIZE | WRITE_SIZE | TCC hit/miss/req | SQ busy/insts  -> gpurun_out/<tag>/pmc_*.csv (kernel, counter, avg per dispatch)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  (cd "$R" && timeout 600 rocprofv3 --pmc $G --output-format csv -d "$OUT/p$i" -- python3 "$@" > "$OUT/p$i.log" 2>&1)
  echo "pass $i ($G): exit $?" >> "$OUT/passes.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, cs in sorted(agg.items()):
        if not any(len(v) >= 3 for v in cs.values()):
            continue
        line = k + " | " + " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items()))
        print(line); fh.write(line + "\n")
PY
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete

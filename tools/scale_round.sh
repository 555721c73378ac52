#!/bin/bash
# The whole multi-GPU measurement in ONE command, for the day an N-GPU MI355X node is available (nothing here has ever run
# over xGMI: the build pool has 1-GPU boxes).  One process per GPU under torch.distributed.run, RCCL (backend "nccl").
#   bash tools/scale_round.sh <tag> [max_gpus=8]
# Writes gpurun_out/<tag>/: one JSON line per run
#   replicas_N.json        bench.py --gpus N                         independent registrations, weak scaling (the headline metric)
#   slab_c3_N.json         bench.py --gpus N --shard slab            ONE C3 batch sharded by z-slab, strong scaling
#   slab_c4_N.json         bench.py --gpus N --shard slab --config c4 --conv-dtype bf16     BASELINE configs[3]
#   train_c5_N.json        tools/train_bench.py --gpus N --config c5 --conv-dtype bf16 --grad-dtype bf16   BASELINE configs[4]
#   train_c3_N.json        tools/train_bench.py --gpus N --config c3 (fp32)
# and summary.txt (value per N; efficiency is for the reader to compute: value_N / (N * value_1) weak, value_N / value_1 / N strong).
set -u
TAG=${1:-scale}
MAXG=${2:-8}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python3 -c "import torch; print(torch.cuda.device_count())")
PORT=29510
run() {   # run <N> <outfile> <program> [args…]
  local n=$1 out=$2; shift 2
  PORT=$((PORT + 1))
  local pick="tail -n 1"                       # bench.py: the JSON line is the last one; train_bench.py: the first one
  case "$out" in train_*) pick="head -n 1";; esac
  if [ "$n" -eq 1 ]; then
    timeout 900 python3 "$@" 2> "$O/$out.err" | $pick > "$O/$out"
  else
    timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $PORT "$@" 2> "$O/$out.err" | $pick > "$O/$out"
  fi
  echo "$out: $(head -c 160 "$O/$out")"
}
for N in 1 2 4 8; do
  [ "$N" -gt "$MAXG" ] && break
  if [ "$N" -gt "$HAVE" ]; then echo "only $HAVE GPU(s) visible: stopping before N=$N"; break; fi
  run $N replicas_$N.json bench.py --gpus $N --no-cpu-baseline --no-drr
  run $N slab_c3_$N.json bench.py --gpus $N --shard slab --no-cpu-baseline
  run $N slab_c4_$N.json bench.py --gpus $N --shard slab --config c4 --conv-dtype bf16 --no-cpu-baseline
  run $N train_c5_$N.json tools/train_bench.py --gpus $N --config c5 --conv-dtype bf16 --grad-dtype bf16
  run $N train_c3_$N.json tools/train_bench.py --gpus $N --config c3
done
python3 - "$O" <<'PY' | tee "$O/summary.txt"
import glob, json, os, re, sys
o = sys.argv[1]
rows = {}
for f in sorted(glob.glob(o + "/*_[0-9].json")):
    m = re.match(r"(.*)_(\d+)\.json", os.path.basename(f))
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        rows.setdefault(m.group(1), {})[int(m.group(2))] = (d.get("value"), d.get("unit"), d.get("ms_per_step"))
    except Exception as e:
        rows.setdefault(m.group(1), {})[int(m.group(2))] = (None, "unreadable: %s" % e, None)
for name, per in rows.items():
    print(name)
    for n in sorted(per):
        v, u, ms = per[n]
        print(f"   N={n}: {v} {u}  ({ms} ms/step)")
PY

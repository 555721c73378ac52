#!/bin/bash
# One gpurun call of the round's standard measurements (development aid):
#   gpurun --timeout 2700 -- 'bash tools/gpu_round.sh r03_final [tests|notests]'
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
if [ "${2:-tests}" = "tests" ]; then
  timeout 1500 python3 -m pytest tests -m gpu -q -x --durations=15 > "$O/pytest.log" 2>&1
  echo "pytest exit $?" >> "$O/pytest.log"
  tail -n 5 "$O/pytest.log"
fi
timeout 500 python3 bench.py 2> "$O/bench.err" | tail -n 1 > "$O/bench.json"
head -c 600 "$O/bench.json"; echo
bash tools/pmc_bench.sh $TAG/pmc > /dev/null 2>&1
tail -n 2 "$O/pmc/summary.txt"
bash tools/pmc_bench.sh $TAG/pmc_c4 --config c4 --conv-dtype bf16 > /dev/null 2>&1
tail -n 2 "$O/pmc_c4/summary.txt"
bash tools/pmc_bench.sh $TAG/pmc_bf16 --conv-dtype bf16 > /dev/null 2>&1
tail -n 2 "$O/pmc_bf16/summary.txt"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/fwd" -- python3 "$R/bench.py" --no-cpu-baseline --no-drr --ramp-seconds 0 --steps 10 --warmup 3 > "$O/fwd.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/fwd_c4" -- python3 "$R/bench.py" --no-cpu-baseline --no-drr --ramp-seconds 0 --steps 10 --warmup 3 --config c4 --conv-dtype bf16 > "$O/fwd_c4.log" 2>&1
find "$O" -name "*kernel_trace.csv" -delete
find "$O" -name "*agent_info.csv" -delete
find "$O" -name "*kernel_stats.csv" | head -n 4
cd "$R"
: > "$O/train_modes.jsonl"
for m in "--config c3" "--config c3 --conv-dtype bf16 --grad-dtype bf16" "--config c5 --conv-dtype bf16 --grad-dtype bf16" "--config c5 --conv-dtype bf16"; do
  timeout 400 python3 tools/train_bench.py $m 2>/dev/null | head -n 1 >> "$O/train_modes.jsonl"
done
timeout 400 python3 tools/train_bench.py --config c3 2>/dev/null > "$O/train_c3_kernels.jsonl"
timeout 400 python3 tools/train_bench.py --config c5 --conv-dtype bf16 --grad-dtype bf16 2>/dev/null > "$O/train_c5_bf16_kernels.jsonl"
timeout 300 python3 bench.py --no-cpu-baseline --no-drr --conv-dtype bf16 2>/dev/null | tail -n 1 > "$O/bench_bf16.json"
timeout 400 python3 bench.py --config c4 --conv-dtype bf16 2>/dev/null | tail -n 1 > "$O/bench_c4_bf16.json"
LIFTREG_CONV0_BF16_PASSES=1 timeout 300 python3 - > "$O/bench_c4_bf16_feature_volume_path.json" 2>/dev/null <<'PY'
import subprocess, sys
# the round-2 path (fp32 feature volume + channel-pass first block): the model's views threshold raised out of reach
code = "import liftreg_amd.models.LiftRegDeformSubspaceBackproj as m; m.model.ENCIN_MIN_VIEWS = 99; import runpy, sys; sys.argv = ['bench.py', '--no-cpu-baseline', '--no-drr', '--config', 'c4', '--conv-dtype', 'bf16']; runpy.run_path('bench.py', run_name='__main__')"
r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
PY
timeout 400 python3 bench.py --no-cpu-baseline --config c4 --conv-dtype bf16 --shard slab 2>/dev/null | tail -n 1 > "$O/bench_c4_bf16_slab_x1.json"
timeout 300 python3 bench.py --no-cpu-baseline --no-drr --shard slab 2>/dev/null | tail -n 1 > "$O/bench_slab_x1.json"
timeout 300 python3 bench.py --no-cpu-baseline --no-drr --config c1 --graph 2>/dev/null | tail -n 1 > "$O/bench_c1_graph.json"
timeout 300 python3 bench.py --no-cpu-baseline --no-drr --config c2 --graph 2>/dev/null | tail -n 1 > "$O/bench_c2_graph.json"
: > "$O/shard_bench.jsonl"
for w in 1 2 4 8; do timeout 120 python3 tools/shard_bench.py --world $w 2>/dev/null | tail -n 1 >> "$O/shard_bench.jsonl"; done
timeout 120 python3 tools/shard_bench.py --world 4 --views 11 --batch 4 --conv-dtype bf16 2>/dev/null | tail -n 1 >> "$O/shard_bench.jsonl"
timeout 120 python3 tools/ab_drr_hu.py > "$O/ab_drr_hu.txt" 2>/dev/null
timeout 120 python3 tools/abl_c0cl.py > "$O/abl_c0cl.txt" 2>/dev/null
timeout 120 python3 tools/ab_encin.py > "$O/ab_encin.txt" 2>/dev/null
timeout 200 python3 tools/ab_conv0_split.py > "$O/ab_conv0_split.txt" 2>/dev/null
timeout 300 python3 bench.py --no-drr --conv0-split 2>/dev/null | tail -n 1 > "$O/bench_conv0_split.json"
bash tools/ab_train_algebra.sh > "$O/ab_train_algebra.txt" 2>/dev/null
bash tools/power_round.sh "$TAG/power" > "$O/power_trace.txt" 2>/dev/null
cat "$O/train_modes.jsonl" "$O/shard_bench.jsonl"
python3 - "$O" <<'PY'
import json, sys, glob, os
o = sys.argv[1]
for f in sorted(glob.glob(o + "/bench*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), round(d["value"], 1), "reg/s", round(d["ms_per_step"], 3), "ms")
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY

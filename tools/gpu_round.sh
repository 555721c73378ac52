#!/bin/bash
# One gpurun call of the round's standard measurements (development aid):
#   gpurun --timeout 2700 -- 'bash tools/gpu_round.sh r02_a [tests|notests]'
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
if [ "${2:-tests}" = "tests" ]; then
  timeout 1800 python3 -m pytest tests -m gpu -q -x --durations=15 > "$O/pytest.log" 2>&1
  echo "pytest exit $?" >> "$O/pytest.log"
  tail -n 5 "$O/pytest.log"
fi
timeout 600 python3 bench.py 2> "$O/bench.err" | tail -n 1 > "$O/bench.json"
cat "$O/bench.json" | head -c 3000; echo
timeout 300 python3 tools/conv0_stamps.py > "$O/conv0_stamps.txt" 2>&1
cat "$O/conv0_stamps.txt"
bash tools/pmc_bench.sh $TAG/pmc > /dev/null 2>&1
tail -n 3 "$O/pmc/summary.txt"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/fwd" -- python3 "$R/bench.py" --no-cpu-baseline --no-drr --ramp-seconds 0 --steps 10 --warmup 3 > "$O/fwd.log" 2>&1
find "$O" -name "*kernel_trace.csv" -delete
find "$O" -name "*agent_info.csv" -delete
find "$O" -name "*kernel_stats.csv" | head -n 2
cd "$R"
: > "$O/train_modes.jsonl"
for m in "--config c3" "--config c3 --conv-dtype bf16 --grad-dtype bf16" "--config c5 --conv-dtype bf16 --grad-dtype bf16"; do
  timeout 600 python3 tools/train_bench.py $m 2>/dev/null | head -n 1 >> "$O/train_modes.jsonl"
done
timeout 600 python3 tools/train_bench.py --config c3 2>/dev/null > "$O/train_c3_kernels.jsonl"
timeout 300 python3 bench.py --no-cpu-baseline --no-drr --conv-dtype bf16 2>/dev/null | tail -n 1 > "$O/bench_bf16.json"
timeout 300 python3 bench.py --no-cpu-baseline --no-drr --config c4 --conv-dtype bf16 2>/dev/null | tail -n 1 > "$O/bench_c4_bf16.json"
timeout 300 python3 bench.py --no-cpu-baseline --no-drr --shard slab 2>/dev/null | tail -n 1 > "$O/bench_slab_x1.json"
timeout 300 python3 bench.py --no-cpu-baseline --no-drr --fuse-bp --fuse-ncc 2>/dev/null | tail -n 1 > "$O/bench_f1_fusions_on.json"
cat "$O/train_modes.jsonl"

#!/bin/bash
# One gpurun call of round 5's standard evidence (development aid):
#   gpurun --timeout 1500 -- 'bash tools/r05_round.sh r05_a'
# -> gpurun_out/<tag>/: bench.json (the default line incl. extra_lines), pmc/{summary.txt,traffic.json} (C3 fp32), pmc_native/ (the
#    reference's shipped 160^3 configuration), fwd/ (rocprofv3 --kernel-trace --stats of the bench step)
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
timeout 600 python3 bench.py 2> "$O/bench.err" | tail -n 1 > "$O/bench.json"
head -c 300 "$O/bench.json"; echo
PMC_DRR=1 bash tools/pmc_bench.sh $TAG/pmc > /dev/null 2>&1
tail -n 2 "$O/pmc/summary.txt"
bash tools/pmc_bench.sh $TAG/pmc_native --config native160 > /dev/null 2>&1
tail -n 2 "$O/pmc_native/summary.txt"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/fwd" -- python3 "$R/bench.py" --no-cpu-baseline --no-drr --ramp-seconds 0 --steps 10 --warmup 3 > "$O/fwd.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/fwd_native" -- python3 "$R/bench.py" --config native160 --no-cpu-baseline --no-drr --ramp-seconds 0 --steps 10 --warmup 3 > "$O/fwd_native.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train" -- python3 "$R/tools/train_bench.py" --config c3 --no-kernel-table > "$O/train.log" 2>&1
find "$O" -name "*kernel_trace.csv" -delete
find "$O" -name "*agent_info.csv" -delete
find "$O" -name "*kernel_stats.csv"
cd "$R"
: > "$O/shard_bench.jsonl"
for w in 1 2 4 8; do timeout 120 python3 tools/shard_bench.py --world $w 2>/dev/null | tail -n 1 >> "$O/shard_bench.jsonl"; done
timeout 120 python3 tools/shard_bench.py --world 4 --views 11 --batch 4 --conv-dtype bf16 2>/dev/null | tail -n 1 >> "$O/shard_bench.jsonl"
timeout 120 python3 tools/shard_bench.py --world 8 --views 11 --batch 4 --conv-dtype bf16 2>/dev/null | tail -n 1 >> "$O/shard_bench.jsonl"
timeout 120 python3 tools/ab_drr.py > "$O/ab_drr.txt" 2>/dev/null
timeout 200 python3 tools/ab_pair01_dense.py --reps 3 > "$O/ab_pair01_dense.txt" 2>/dev/null
timeout 300 python3 bench.py --config native160 2>/dev/null | tail -n 1 > "$O/bench_native160.json"
timeout 300 python3 tools/train_bench.py --config c3 2>/dev/null > "$O/train_c3_kernels.jsonl"
cat "$O/shard_bench.jsonl"

#!/bin/bash
# One gpurun call of round 6's standard evidence (development aid):
#   gpurun --timeout 2400 -- 'bash tools/r06_round.sh r06_a'
# -> gpurun_out/<tag>/: bench.out (the driver's command: '#full' lines + the final line), pmc/{summary.txt,traffic.json} (C3 fp32),
#    pmc_native/ (the reference's shipped 160^3 configuration), fwd*/ train*/ (rocprofv3 --kernel-trace --stats), shard_bench.jsonl, A/B texts
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > "$O/bench.out" 2> "$O/bench.err"
tail -n 1 "$O/bench.out" | head -c 400; echo; tail -n 4 "$O/bench.err"
PMC_DRR=1 bash tools/pmc_bench.sh $TAG/pmc > /dev/null 2>&1
tail -n 2 "$O/pmc/summary.txt"
bash tools/pmc_bench.sh $TAG/pmc_native --config native160 > /dev/null 2>&1
tail -n 2 "$O/pmc_native/summary.txt"
cd /tmp && export TMPDIR=/tmp
prof() { d=$1; shift; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$d" -- python3 "$@" > "$O/$d.log" 2>&1; }
prof fwd "$R/bench.py" --no-cpu-baseline --no-drr --ramp-seconds 0 --steps 18 --warmup 3 --extra-lines off
prof fwd_native "$R/bench.py" --config native160 --no-cpu-baseline --no-drr --ramp-seconds 0 --steps 10 --warmup 3 --extra-lines off
prof train_c3 "$R/tools/train_bench.py" --config c3 --no-kernel-table --ramp-seconds 0 --steps 7 --warmup 2
prof train_native160 "$R/tools/train_bench.py" --config native160 --no-kernel-table --ramp-seconds 0 --steps 7 --warmup 2
prof train_c5_bf16 "$R/tools/train_bench.py" --config c5 --conv-dtype bf16 --grad-dtype bf16 --no-kernel-table --ramp-seconds 0 --steps 7 --warmup 2
find "$O" -name "*kernel_trace.csv" -delete
find "$O" -name "*agent_info.csv" -delete
find "$O" -name "*kernel_stats.csv"
cd "$R"
: > "$O/shard_bench.jsonl"
for w in 1 2 4 8; do timeout 200 python3 tools/shard_bench.py --world $w --graph 2>/dev/null | tail -n 1 >> "$O/shard_bench.jsonl"; done
timeout 200 python3 tools/shard_bench.py --world 4 --views 11 --batch 4 --conv-dtype bf16 --graph 2>/dev/null | tail -n 1 >> "$O/shard_bench.jsonl"
timeout 200 python3 tools/shard_bench.py --world 8 --views 11 --batch 4 --conv-dtype bf16 --graph 2>/dev/null | tail -n 1 >> "$O/shard_bench.jsonl"
timeout 300 python3 tools/shard_bench.py --procs 4 --backend auto 2>/dev/null | tail -n 6 > "$O/shard_bench_procs4.jsonl"
timeout 120 python3 tools/ab_drr.py > "$O/ab_drr.txt" 2>/dev/null
timeout 120 python3 tools/ab_backproject.py > "$O/ab_backproject.txt" 2>/dev/null
timeout 200 python3 tools/ab_fused_bwd.py > "$O/ab_fused_bwd.txt" 2>/dev/null
timeout 100 ./tools/micro/lds_unaligned > "$O/lds_unaligned.txt" 2>/dev/null
timeout 300 python3 tools/train_bench.py --config c3 2>/dev/null > "$O/train_c3_kernels.jsonl"
timeout 300 python3 tools/train_bench.py --config native160 2>/dev/null > "$O/train_native160_kernels.jsonl"
timeout 300 python3 tools/train_bench.py --config c5 --conv-dtype bf16 --grad-dtype bf16 --vs-fp32 cpu 2>/dev/null > "$O/train_c5_bf16_kernels.jsonl"
cat "$O/shard_bench.jsonl"

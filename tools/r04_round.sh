#!/bin/bash
# One gpurun call of round 4's standard measurements (development aid):
#   gpurun --timeout 2400 -- 'bash tools/r04_round.sh r04_a [tests|notests]'
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
if [ "${2:-tests}" = "tests" ]; then
  timeout 1500 python3 -m pytest tests -m gpu -q --durations=10 > "$O/pytest.log" 2>&1
  echo "pytest exit $?" >> "$O/pytest.log"
  tail -n 3 "$O/pytest.log"
fi
timeout 500 python3 bench.py 2> "$O/bench.err" | tail -n 1 > "$O/bench.json"
# interleaved A/B of the whole step: the two first blocks as two fp32-MFMA kernels (round 3) | as the fused pair kernel
: > "$O/ab_pair01_step.txt"
for i in 1 2 3; do
  for a in "--no-pair01" ""; do
    timeout 300 python3 bench.py --no-cpu-baseline --no-drr $a 2>/dev/null | tail -n 1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$a' or 'pair01', round(r['value'],1), 'reg/s', round(r['ms_per_step'],3), 'ms  sclk', r['roofline'].get('sclk_mhz'), {k:v['ms'] for k,v in r['kernels'].items() if v['ms']>0.1})" >> "$O/ab_pair01_step.txt"
  done
done
cat "$O/ab_pair01_step.txt"
bash tools/pmc_bench.sh $TAG/pmc > /dev/null 2>&1
tail -n 1 "$O/pmc/summary.txt"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/fwd" -- python3 "$R/bench.py" --no-cpu-baseline --no-drr --ramp-seconds 0 --steps 10 --warmup 3 > "$O/fwd.log" 2>&1
find "$O" -name "*kernel_trace.csv" -delete
find "$O" -name "*agent_info.csv" -delete
cd "$R"
timeout 300 python3 bench.py --no-cpu-baseline --no-drr --shard slab 2>/dev/null | tail -n 1 > "$O/bench_slab_x1.json"
: > "$O/shard_bench.jsonl"
for w in 1 2 4 8; do timeout 120 python3 tools/shard_bench.py --world $w 2>/dev/null | tail -n 1 >> "$O/shard_bench.jsonl"; done
cat "$O/shard_bench.jsonl"
timeout 300 python3 bench.py --no-cpu-baseline --no-drr --conv-dtype bf16 2>/dev/null | tail -n 1 > "$O/bench_bf16.json"
timeout 400 python3 bench.py --no-cpu-baseline --config c4 --conv-dtype bf16 2>/dev/null | tail -n 1 > "$O/bench_c4_bf16.json"
python3 - "$O" <<'PY'
import json, sys, glob, os
o = sys.argv[1]
for f in sorted(glob.glob(o + "/bench*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), round(d["value"], 1), "reg/s", round(d["ms_per_step"], 3), "ms")
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
# bf16 z-march A/B, training A/B, pipelines (round 4, second half)
bash tools/ab_bf16_march.sh $TAG/bf16_march > /dev/null 2>&1
cp "$O/bf16_march/ab.txt" "$O/ab_bf16_march.txt" 2>/dev/null
bash tools/ab_pair01_train.sh > "$O/ab_pair01_train.txt" 2>&1
for s in 1 3 1 3; do timeout 300 python3 bench.py --no-cpu-baseline --no-drr --streams $s 2>/dev/null | tail -n 1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('streams', r['config'].get('streams'), round(r['value'],1), 'reg/s', round(r['ms_per_step'],3), 'ms')" >> "$O/ab_streams.txt"; done
bash tools/pmc_bench.sh $TAG/pmc_bf16 --conv-dtype bf16 > /dev/null 2>&1
bash tools/pmc_bench.sh $TAG/pmc_c4_bf16 --config c4 --conv-dtype bf16 > /dev/null 2>&1
cat "$O/ab_bf16_march.txt" "$O/ab_pair01_train.txt" "$O/ab_streams.txt"

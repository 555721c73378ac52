#!/bin/bash
# Re-measure everything profiles/ holds, in one gpurun call (development aid).
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r01_i'
# Writes gpurun_out/<tag>/…; copy what should be judged into profiles/ afterwards (tools/refresh_profiles.sh does not
# touch profiles/).  rocprofv3 runs the program itself (python3 …), tracing and PMC passes are separate runs.
set -u
TAG=${1:-refresh}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
last() { tail -n 1; }
python3 $R/bench.py 2>$O/l_bench.log | last > $O/bench.json
python3 $R/bench.py --no-cpu-baseline --conv-dtype bf16 2>/dev/null | last > $O/bench_bf16.json
python3 $R/bench.py --no-cpu-baseline --conv-dtype bf16 --pca-dtype bf16 2>/dev/null | last > $O/bench_bf16_pcabf16.json
python3 $R/bench.py --no-cpu-baseline --config c4 --conv-dtype bf16 2>/dev/null | last > $O/bench_c4_bf16.json
python3 $R/bench.py --no-cpu-baseline --config c1 --graph 2>/dev/null | last > $O/bench_c1_graph.json
python3 $R/bench.py --no-cpu-baseline --config c2 --graph 2>/dev/null | last > $O/bench_c2_graph.json
: > $O/train_modes.jsonl
for m in "--config c3" "--config c5" "--config c3 --conv-dtype bf16" "--config c3 --conv-dtype bf16 --grad-dtype bf16" \
         "--config c3 --conv-dtype bf16 --grad-dtype bf16 --pca-dtype bf16" "--config c5 --conv-dtype bf16" \
         "--config c5 --conv-dtype bf16 --grad-dtype bf16"; do
  python3 $R/tools/train_bench.py $m 2>/dev/null | head -n 1 >> $O/train_modes.jsonl
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/l_fwd.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16 -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 --conv-dtype bf16 > $O/l_bf16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/tools/train_bench.py --config c3 --steps 5 > $O/l_train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trainbf -- python3 $R/tools/train_bench.py --config c3 --steps 5 --conv-dtype bf16 --grad-dtype bf16 > $O/l_trainbf.log 2>&1
find $O -name "*kernel_stats.csv" | head
# raw traces are large: keep the stats only
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete

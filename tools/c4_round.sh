#!/bin/bash
# C4 development round: the full-size C4 tests + the bf16 tests, then the C4 bf16 bench line (old channel-pass kernel for A/B)
TAG=${1:-r03_c4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG; mkdir -p "$O"; cd "$R"
timeout 900 python3 -m pytest tests/test_gpu_c4_full.py tests/test_gpu_bf16.py tests/test_gpu_sharded_model.py tests/test_gpu_fuzz.py -m gpu -q -x > "$O/pytest.log" 2>&1; echo "pytest exit $?" >> "$O/pytest.log"; tail -n 6 "$O/pytest.log"
timeout 300 python3 bench.py --no-cpu-baseline --no-drr --config c4 --conv-dtype bf16 2>"$O/bench.err" | tail -n 1 > "$O/bench_c4_bf16.json"
LIFTREG_CONV0_BF16_PASSES=1 timeout 300 python3 bench.py --no-cpu-baseline --no-drr --config c4 --conv-dtype bf16 2>/dev/null | tail -n 1 > "$O/bench_c4_bf16_old.json"
python3 - <<PY
import json
for f in ("bench_c4_bf16","bench_c4_bf16_old"):
    d=json.load(open("$O/"+f+".json")); print(f, round(d["value"],1), "reg/s", {k:v["ms"] for k,v in d["kernels"].items() if v["ms"]>0.05})
PY

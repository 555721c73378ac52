#!/bin/bash
# A/B of the bf16 16->32 block: z-marching kernel (default) | row kernel (LIFTREG_BF16_NO_MARCH=1); development aid.
#   gpurun --timeout 1200 -- 'bash tools/ab_bf16_march.sh r04_bf16 [tests]'
set -u
TAG=${1:-r04_bf16}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
if [ "${2:-}" = "tests" ]; then
  timeout 900 python3 -m pytest tests/test_gpu_bf16.py tests/test_gpu_c4_full.py tests/test_gpu_sharded_model.py -m gpu -q -x > "$O/pytest.log" 2>&1
  echo "pytest exit $?" >> "$O/pytest.log"
  tail -n 5 "$O/pytest.log"
fi
: > "$O/ab.txt"
for i in 1 2; do
  for cfg in c3 c4; do
    for a in 1 0; do
      LIFTREG_BF16_NO_MARCH=$a timeout 300 python3 bench.py --no-cpu-baseline --no-drr --config $cfg --conv-dtype bf16 2>/dev/null | tail -n 1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$cfg no_march=$a', round(r['value'],1), 'reg/s', round(r['ms_per_step'],3), 'ms', {k:v['ms'] for k,v in r['kernels'].items() if 'conv' in k})" >> "$O/ab.txt"
    done
  done
done
cat "$O/ab.txt"

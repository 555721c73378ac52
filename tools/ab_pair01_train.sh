#!/bin/bash
# Training step (fp32 C3) with the forward of blocks 0 + 1 through the fused pair kernel (default) | the two fp32-MFMA kernels
# (--model-opt fuse_pair01_train=false), interleaved on one box (development aid).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for i in 1 2 3; do
  for opt in "" "--model-opt fuse_pair01_train=false"; do
    timeout 300 python3 tools/train_bench.py --config c3 --steps 10 --warmup 3 $opt 2>/dev/null | python3 -c "
import sys, json
rows = [json.loads(l) for l in sys.stdin if l.startswith('{')]
head = [r for r in rows if 'ms_per_train_step' in r][0]
top = {r['kernel']: r['ms'] for r in rows if 'kernel' in r and r['ms'] * r['launches'] / 13 > 0.5}
print('${opt:-pair01_train}', head['ms_per_train_step'], 'ms', top)"
  done
done

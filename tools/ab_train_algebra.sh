#!/bin/bash
# The two algebraic changes of the training step, switched off one at a time, on ONE box (development aid):
#   regulariser on the PCA coefficients (reg_in_coef_space), similarity gradient through its moments (ncc_grad_via_moments)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
echo "# tools/ab_train_algebra.sh on one MI355X: training step (ms) with the two algebraic routes switched off one at a time"
echo "# (4th row of each block = both off and torch's default multi-pass Adam; first and last row = default, twice)"
for cfg in "--config c3" "--config c3 --conv-dtype bf16 --grad-dtype bf16" "--config c5 --conv-dtype bf16 --grad-dtype bf16"; do
  echo "# $cfg"
  for opt in "" "--model-opt reg_in_coef_space=false" "--model-opt ncc_grad_via_moments=false" "--model-opt reg_in_coef_space=false --model-opt ncc_grad_via_moments=false --adam-foreach" ""; do
    ms=$(timeout 300 python3 tools/train_bench.py $cfg $opt --no-kernel-table 2>/dev/null | tail -n 1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_train_step'])")
    echo "${opt:-default}: $ms ms"
  done
done

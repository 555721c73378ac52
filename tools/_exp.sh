cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_training.py tests/test_gpu_fuzz.py tests/test_gpu_bf16.py -m gpu -q -x 2>&1 | tail -n 2
python3 tools/train_bench.py --config c3 --steps 5 2>/dev/null | head -n 1 | cut -c1-200
python3 tools/train_bench.py --config c3 --steps 5 2>/dev/null | grep "wgrad_c"
LIFTREG_WGRAD_ROWS=1 python3 tools/train_bench.py --config c3 --steps 5 2>/dev/null | grep "wgrad_c16x32_256\|wgrad_c32x32_128"

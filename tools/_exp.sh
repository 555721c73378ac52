cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -q 2>&1 | tail -n 4
LIFTREG_FUZZ_CASES=300 LIFTREG_FUZZ_SEED=7 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "round2 or conv_forward or whole_model" 2>&1 | tail -n 3
LIFTREG_FUZZ_CASES=100 LIFTREG_FUZZ_LONG_H=520 LIFTREG_FUZZ_SEED=8 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "round2 or conv_forward" 2>&1 | tail -n 3
python3 __graft_entry__.py smoke 2>&1 | tail -n 1

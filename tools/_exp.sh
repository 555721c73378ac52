cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_headline.py tests/test_gpu_sharded_model.py -m gpu -q -x -k "ncc or slab or sharded" 2>&1 | tail -n 3
python3 tools/abdecode.py 2>&1 | tail -n 2

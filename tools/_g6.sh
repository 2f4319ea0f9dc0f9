cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_k2.py -x -q -k "sparse_tiles_equal" 2>&1 | tail -30

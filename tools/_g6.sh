cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_mgpu.py -x -q -k "ranked_rungs" 2>&1 | tail -30

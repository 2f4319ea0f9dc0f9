cd $GRAFT_REPO_ROOT
D2G_BS_SPARSE_MIN_N=1 D2G_SP_TILE_FRAC=1 python -m pytest tests/test_gpu_k2.py -q -m gpu -k fill_ahead 2>&1 | tail -2
python -m pytest tests/test_gpu_k2.py -q -m gpu -k fill_ahead 2>&1 | tail -2

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c27
timeout 1500 python -m pytest tests/test_gpu_k2.py -q -m gpu -x -k "fill_ahead" > gpurun_out/c27/k2.log 2>&1; tail -30 gpurun_out/c27/k2.log

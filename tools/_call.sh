cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c10
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/c10/gpu_all.log 2>&1; tail -6 gpurun_out/c10/gpu_all.log
D2G_BS_SPARSE_MIN_N=1 D2G_SP_TILE_FRAC=1 timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/c10/gpu_all_forced.log 2>&1; tail -6 gpurun_out/c10/gpu_all_forced.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/c10/bench.out 2> gpurun_out/c10/bench.err; tail -c 1500 gpurun_out/c10/bench.out; tail -3 gpurun_out/c10/bench.err
tools/mgpu_model.sh > gpurun_out/c10/mm.log 2>&1; tail -12 gpurun_out/c10/mm.log | cut -c1-700

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c18
timeout 1200 python -m pytest tests/test_gpu_k2.py tests/test_gpu_mgpu.py -q -m gpu -x > gpurun_out/c18/k2.log 2>&1; tail -4 gpurun_out/c18/k2.log
(for m in "" "MATRIX=noise C=1" "MATRIX=noise C=3" "N=50000"; do echo -n "$m: "; env $m timeout 300 python tools/k2_time.py 2>&1 | grep step | cut -c1-200; done) | tee gpurun_out/c18/times.txt
tools/kstats.sh c18 python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "emit\|sparse_k\|place" gpurun_out/c18_kernel_stats.txt
N=50000 tools/kstats.sh c18_50k python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; grep "emit\|sparse_k\|place" gpurun_out/c18_50k_kernel_stats.txt

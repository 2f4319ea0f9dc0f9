cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c15
timeout 1500 python -m pytest tests/test_gpu_k2.py tests/test_gpu_mgpu.py -q -m gpu -x > gpurun_out/c15/k2.log 2>&1; tail -6 gpurun_out/c15/k2.log
(for m in stated noise; do echo -n "$m: "; MATRIX=$m C=1 timeout 120 python tools/k2_time.py 2>&1 | grep step | cut -c1-330; done
echo -n "stated cert=0: "; D2G_SP_CERT=0 timeout 120 python tools/k2_time.py 2>&1 | grep step | cut -c1-330
echo -n "N=50000: "; N=50000 MATRIX=noise C=0 timeout 300 python tools/k2_time.py 2>&1 | grep step| cut -c1-330
) > gpurun_out/c15/k2_times.txt 2>&1; cat gpurun_out/c15/k2_times.txt
tools/kstats.sh c15_stated python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1; head -20 gpurun_out/c15_stated_kernel_stats.txt
timeout 300 python tools/plist_stats.py > gpurun_out/c15/plist_stats.txt 2>&1; cat gpurun_out/c15/plist_stats.txt | cut -c1-700

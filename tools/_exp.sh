cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_k2.py tests/test_gpu_mgpu.py -x -q > gpurun_out/exp6_tests.log 2>&1; tail -3 gpurun_out/exp6_tests.log
rm -f gpurun_out/exp6_times.txt
for m in 0 1; do D2G_SP_PERMUTE_LDS=$m STEPS=200 python tools/k2_time.py 2>/dev/null | grep step | cut -c1-70 | sed "s/^/lds=$m /"; done >> gpurun_out/exp6_times.txt
for n in 18000 30000 50000; do for m in 0 1; do N=$n D2G_SP_PERMUTE_LDS=$m STEPS=40 python tools/k2_time.py 2>/dev/null | grep step | cut -c1-70 | sed "s/^/N=$n lds=$m /"; done; done >> gpurun_out/exp6_times.txt
cat gpurun_out/exp6_times.txt
N=50000 STEPS=10 tools/kstats.sh exp6_50k python $GRAFT_REPO_ROOT/tools/k2_time.py | grep -E "permute|planes|kernel  "
N=18000 STEPS=10 tools/kstats.sh exp6_18k python $GRAFT_REPO_ROOT/tools/k2_time.py | grep -E "permute|planes|kernel  "
N=18000 D2G_SP_PERMUTE_LDS=0 STEPS=10 tools/kstats.sh exp6_18k0 python $GRAFT_REPO_ROOT/tools/k2_time.py | grep -E "permute|planes|kernel  "

# sub-tiles one column word wide (SP_JR = 1) + first look on a side stream (default) vs inline (variant)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_k2.py -x -q -m gpu 2>&1 | tail -3
for c in 0 10; do if [ $c = 0 ]; then M=stated; else M=noise; fi; MATRIX=$M C=$c STEPS=20 timeout 300 tools/kstats.sh p15_c$c python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "k2_bits|sp_list|sp_permute" gpurun_out/p15_c${c}_kernel_stats.txt | cut -c1-60,93-150; done
for c in 0 1 3 10; do if [ $c = 0 ]; then M=stated; else M=noise; fi; echo -n "c=$c "; MATRIX=$M C=$c timeout 200 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50; done
N=50000 MATRIX=stated STEPS=5 timeout 300 tools/kstats.sh p15_50k python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "k2_bits" gpurun_out/p15_50k_kernel_stats.txt | cut -c1-60,93-150
N=50000 MATRIX=stated STEPS=5 timeout 300 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50
for lib in "" $R/dashing2_amd/libd2g_sampinline.so; do echo "lib=$lib"; for m in "noise 100" "skewed -" "stated -" "unrelated -"; do set -- $m; D2G_LIB=$lib MATRIX=$1 C=$2 timeout 120 python3 tools/k2_first.py 2>&1 | grep "first step" | cut -c1-150; done; done
for m in "skewed -" "unrelated -"; do set -- $m; D2G_BS_SPARSE=0 MATRIX=$1 C=$2 timeout 120 python3 tools/k2_first.py 2>&1 | grep "first step" | cut -c1-150; done

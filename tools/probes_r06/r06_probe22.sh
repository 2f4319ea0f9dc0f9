cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_k2.py -x -q -m gpu 2>&1 | grep -a -v "^Extension modules\|^  File" | tail -3
D2G_LIB=$R/dashing2_amd/libd2g_trace.so C=10 timeout 200 python3 tools/emit_trace.py 2>&1 | grep -v amdgpu.ids
for c in 0 1 3 10; do if [ $c = 0 ]; then M=stated; else M=noise; fi; echo -n "c=$c "; MATRIX=$M C=$c timeout 200 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50; done
for c in 1 10; do MATRIX=noise C=$c STEPS=20 timeout 300 tools/kstats.sh p22_c$c python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "sp_emit|sp_pairs" gpurun_out/p22_c${c}_kernel_stats.txt | cut -c1-40,93-150; done
N=50000 MATRIX=noise C=10 STEPS=5 timeout 300 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50

# batched loads in the count / compose scans + the rank kernel's own-value mask: parity, then kernel table at c = 10 and c = 1, N = 50000 c = 10
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_k2.py -x -q -m gpu 2>&1 | tail -3
for c in 10; do MATRIX=noise C=$c STEPS=20 timeout 300 tools/kstats.sh p12_c$c python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "sp_|k2_bits|bs_" gpurun_out/p12_c${c}_kernel_stats.txt | cut -c1-60,93-150; done
for c in 0 1 3 10; do if [ $c = 0 ]; then M=stated; else M=noise; fi; echo -n "c=$c "; MATRIX=$M C=$c timeout 200 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50; done
N=50000 MATRIX=noise C=10 STEPS=5 timeout 300 tools/kstats.sh p12_50k python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "sp_|k2_bits|bs_" gpurun_out/p12_50k_kernel_stats.txt | cut -c1-60,93-150
N=50000 MATRIX=noise C=10 STEPS=5 timeout 300 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50

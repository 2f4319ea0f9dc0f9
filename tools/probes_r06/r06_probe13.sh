# emit pass A with runs of one (value, segment) aggregated; sparse pair kernel without lane carried through the walk
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_k2.py -x -q -m gpu 2>&1 | tail -3
for c in 0 10; do if [ $c = 0 ]; then M=stated; else M=noise; fi; MATRIX=$M C=$c STEPS=20 timeout 300 tools/kstats.sh p13_c$c python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "sp_emit|sp_pairs|k2_bits|sp_compose|sp_bin|sp_permute" gpurun_out/p13_c${c}_kernel_stats.txt | cut -c1-60,93-150; done
for c in 0 1 3 10; do if [ $c = 0 ]; then M=stated; else M=noise; fi; echo -n "c=$c "; MATRIX=$M C=$c timeout 200 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50; done
N=50000 MATRIX=stated STEPS=5 timeout 300 tools/kstats.sh p13_50k python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "sp_emit|k2_bits" gpurun_out/p13_50k_kernel_stats.txt | cut -c1-60,93-150

# emit kernel with 1024 threads and 2560 values per step (one step per column at config 3) vs 512 threads / 1536 values
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
for v in "" emit1024; do
  lib=""; [ -n "$v" ] && lib=$R/dashing2_amd/libd2g_$v.so
  echo "== variant [$v]"
  D2G_LIB=$lib timeout 600 python -m pytest tests/test_gpu_k2.py -x -q -m gpu 2>&1 | grep -a -v "^Extension modules\|^  File" | tail -2
  for c in 0 10; do if [ $c = 0 ]; then M=stated; else M=noise; fi; D2G_LIB=$lib MATRIX=$M C=$c STEPS=20 timeout 300 tools/kstats.sh p17_${v}_c$c python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "sp_emit|k2_bitslice_sparse" gpurun_out/p17_${v}_c${c}_kernel_stats.txt | cut -c1-40,93-150; done
  for c in 0 1 3 10; do if [ $c = 0 ]; then M=stated; else M=noise; fi; echo -n "c=$c "; D2G_LIB=$lib MATRIX=$M C=$c timeout 200 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50; done
  D2G_LIB=$lib N=50000 MATRIX=stated STEPS=5 timeout 300 tools/kstats.sh p17_50k_$v python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "sp_emit" gpurun_out/p17_50k_${v}_kernel_stats.txt | cut -c1-40,93-150
done

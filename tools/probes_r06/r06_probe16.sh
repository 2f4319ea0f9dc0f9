# waves per sub-tile (D2G_SP_KS 4 / 8 / 16) x sub-tile width (SP_JR 1 / 2): the pair kernel's walk is one memory round trip per plane per wave
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_k2.py -x -q -m gpu 2>&1 | grep -a -v "^Extension modules\|^  File" | tail -3
for v in "" ks8 ks16 ks8jr2 jr2; do
  lib=""; [ -n "$v" ] && lib=$R/dashing2_amd/libd2g_$v.so
  echo "== variant [$v]"
  D2G_LIB=$lib MATRIX=stated STEPS=20 timeout 300 tools/kstats.sh p16_$v python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "k2_bitslice_sparse" gpurun_out/p16_${v}_kernel_stats.txt | cut -c1-40,93-150
  D2G_LIB=$lib N=50000 MATRIX=stated STEPS=5 timeout 300 tools/kstats.sh p16_50k_$v python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "k2_bitslice_sparse|sp_side|bs_rank|sp_scan|sp_place" gpurun_out/p16_50k_${v}_kernel_stats.txt | cut -c1-40,93-150
  D2G_LIB=$lib MATRIX=stated timeout 200 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50
  D2G_LIB=$lib N=50000 MATRIX=stated STEPS=5 timeout 200 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50
done

# pair kernel: mismatch counts accumulated in LDS per group (42 VGPRs), 8 waves per sub-tile (default) vs 4 / 16
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_k2.py -x -q -m gpu 2>&1 | grep -a -v "^Extension modules\|^  File" | tail -2
D2G_LIB=$R/dashing2_amd/libd2g_trace.so timeout 200 python3 tools/sp_trace.py 2>&1 | grep -v amdgpu.ids | head -9 | cut -c1-140
for v in "" ks4 ks16; do
  lib=""; [ -n "$v" ] && lib=$R/dashing2_amd/libd2g_$v.so
  echo "== variant [$v]"
  D2G_LIB=$lib MATRIX=stated STEPS=20 timeout 300 tools/kstats.sh p23_$v python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "k2_bitslice_sparse" gpurun_out/p23_${v}_kernel_stats.txt | cut -c1-40,93-150
  D2G_LIB=$lib N=50000 MATRIX=stated STEPS=5 timeout 300 tools/kstats.sh p23_50k_$v python3 $R/tools/k2_time.py > /dev/null 2>&1; grep -E "k2_bitslice_sparse" gpurun_out/p23_50k_${v}_kernel_stats.txt | cut -c1-40,93-150
  D2G_LIB=$lib MATRIX=stated timeout 200 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50
  D2G_LIB=$lib MATRIX=noise C=10 timeout 200 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50
  D2G_LIB=$lib N=50000 MATRIX=stated STEPS=5 timeout 200 python3 tools/k2_time.py 2>/dev/null | grep "^step" | cut -c1-50
done

cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
for v in "" samp8; do
  lib=""; [ -n "$v" ] && lib=$R/dashing2_amd/libd2g_$v.so
  echo "== variant [$v]"
  D2G_LIB=$lib timeout 600 python -m pytest tests/test_gpu_k2.py -x -q -m gpu -k "first_look or give_up or bench_matrix" 2>&1 | grep -a -v "^Extension modules\|^  File" | tail -2
  for m in "noise 100" "skewed -" "unrelated -" "paired -" "noise 10"; do set -- $m; D2G_LIB=$lib MATRIX=$1 C=$2 timeout 120 python3 tools/k2_first.py 2>&1 | grep "first step" | cut -c1-150; done
done
for m in "noise 100" "skewed -" "unrelated -" "paired -" "noise 10"; do set -- $m; D2G_BS_SPARSE=0 MATRIX=$1 C=$2 timeout 120 python3 tools/k2_first.py 2>&1 | grep "first step" | cut -c1-150; done

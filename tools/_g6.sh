cd $GRAFT_REPO_ROOT
for n in 2048 4096 8192 16384 25000; do
  for sp in 1 0; do echo -n "N=$n sparse=$sp "; N=$n D2G_BS_SPARSE=$sp D2G_BS_SPARSE_MIN_N=1 MATRIX=stated python tools/k2_time.py 2>/dev/null | grep step | cut -c1-42; done
done

cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_k2.py -x -q 2>&1 | tail -2
D2G_BS_SPARSE_MIN_N=1 timeout 1500 python -m pytest tests/test_gpu_k2.py -x -q 2>&1 | tail -2
for m in stated unrelated paired skewed; do for sp in 1 0; do echo -n "N=10000 $m sparse=$sp: "; D2G_BS_SPARSE=$sp MATRIX=$m python tools/k2_time.py 2>/dev/null | grep step | cut -c1-44; done; done
echo -n "N=50000: "; N=50000 MATRIX=stated python tools/k2_time.py 2>/dev/null | grep step | cut -c1-44

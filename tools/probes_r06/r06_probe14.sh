# first look polled instead of synchronised, enqueued behind the rank kernel: first step vs steady step, sparse on/off
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_k2.py -x -q -m gpu 2>&1 | tail -3
for m in "noise 1" "noise 10" "noise 100" "paired -" "skewed -" "stated -" "unrelated -"; do set -- $m; for sp in 1 0; do D2G_BS_SPARSE=$sp MATRIX=$1 C=$2 timeout 120 python3 tools/k2_first.py 2>&1 | grep "first step" | cut -c1-150; done; done

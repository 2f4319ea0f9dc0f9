cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_k2.py -x -q -m gpu -k "first_look or give_up or bench_matrix" 2>&1 | grep -a -v "^Extension modules\|^  File" | tail -4
for m in "unrelated -" "skewed -" "stated -" "noise 1" "noise 10" "noise 30" "paired -"; do set -- $m; MATRIX=$1 C=$2 timeout 120 python3 tools/k2_first.py 2>&1 | grep "first step" | cut -c1-330; done
for n in 2000 50000; do for m in "unrelated -" "stated -" "noise 10"; do set -- $m; N=$n MATRIX=$1 C=$2 REPS=5 timeout 300 python3 tools/k2_first.py 2>&1 | grep "first step" | cut -c1-330; done; done

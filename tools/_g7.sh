cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
echo "== forced sparse everywhere"
D2G_BS_SPARSE_MIN_N=1 timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r04/bench_n1_b.json 2> gpurun_out/r04/bench_n1_b.err; tail -4 gpurun_out/r04/bench_n1_b.err

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
tools/kstats.sh r04_sparse_c3 python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-config4 --no-matrices --no-cpu-baseline --no-sketch --no-multiset --no-traffic > /dev/null 2>&1
cat gpurun_out/r04_sparse_c3_kernel_stats.txt | cut -c1-150
tools/kstats.sh r04_sparse_c4 python $GRAFT_REPO_ROOT/bench.py --sketches 50000 --steps 5 --warmup 2 --no-config4 --no-matrices --no-cpu-baseline --no-sketch --no-multiset --no-traffic > /dev/null 2>&1
cat gpurun_out/r04_sparse_c4_kernel_stats.txt | cut -c1-150

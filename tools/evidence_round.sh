# round-4 evidence on the final commit (GPU box): everything lands in gpurun_out/, the summaries are copied to profiles/ afterwards
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r04_gputests.log 2>&1; tail -3 gpurun_out/r04_gputests.log
D2G_BS_SPARSE_MIN_N=1 D2G_SP_SEG_DIV=1 python -m pytest tests -q -m gpu > gpurun_out/r04_gputests_sparse_forced.log 2>&1; tail -3 gpurun_out/r04_gputests_sparse_forced.log
tools/pmc_round.sh > gpurun_out/pmc_round.log 2>&1; tail -3 gpurun_out/pmc_round.log
cp gpurun_out/r04_pmc.json profiles/r04_pmc.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench.out 2> gpurun_out/r04_bench.err; tail -1 gpurun_out/r04_bench.out > gpurun_out/r04_bench.json
D2G_BS_SPARSE=0 python bench.py --steps 20 --warmup 5 --no-sketch --no-multiset --no-cpu-baseline > gpurun_out/r04_bench_dense.out 2>/dev/null; tail -1 gpurun_out/r04_bench_dense.out > gpurun_out/r04_bench_dense_walk.json
tools/kstats.sh r04_bench python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-config4 --no-matrices --no-cpu-baseline --no-sketch --no-multiset --no-traffic > /dev/null 2>&1
tools/kstats.sh r04_bench_all_legs python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > /dev/null 2>&1
D2G_BS_SPARSE=0 tools/kstats.sh r04_bench_dense_walk python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-config4 --no-matrices --no-cpu-baseline --no-sketch --no-multiset --no-traffic > /dev/null 2>&1
for m in stated unrelated paired skewed; do for sp in 1 0; do echo -n "N=10000 $m sparse=$sp: "; D2G_BS_SPARSE=$sp MATRIX=$m python tools/k2_time.py 2>/dev/null | grep step; done; done > gpurun_out/r04_k2_matrices.txt
for n in 10000 50000; do echo -n "N=$n stated sparse=1 D2G_SP_SEGMENTS=0 (tiles always marked): "; N=$n D2G_SP_SEGMENTS=0 MATRIX=stated python tools/k2_time.py 2>/dev/null | grep step; done >> gpurun_out/r04_k2_matrices.txt
for sp in 1 0; do echo -n "N=50000 stated sparse=$sp: "; N=50000 D2G_BS_SPARSE=$sp MATRIX=stated python tools/k2_time.py 2>/dev/null | grep step; done >> gpurun_out/r04_k2_matrices.txt
N=50000 MATRIX=stated tools/kstats.sh r04_k2_config4 python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1
( time python bench.py --gpus 8 --loopback --steps 5 --warmup 2 ) > gpurun_out/r04_bench_w8_loopback.json 2> gpurun_out/r04_bench_w8_loopback.err
tools/mgpu_model.sh > gpurun_out/mm.log 2>&1; tail -5 gpurun_out/mm.log
D2G_VERBOSE_EXIT=1 python tools/e2e_cli.py --genomes 1000 --threads 112 --big-sketches 50000 > gpurun_out/r04_e2e_cli.txt 2>&1
python tools/cmp_setup_time.py > gpurun_out/r04_cmp_setup_time.txt 2>&1; for v in A B C D; do python tools/cmp_setup_time2.py $v; done >> gpurun_out/r04_cmp_setup_time.txt 2>&1
python tools/k0_time.py 200 5 > gpurun_out/r04_k0_time.txt 2>&1
python tools/fuzz_parity.py 600 404 > gpurun_out/r04_fuzz.txt 2>&1; tail -2 gpurun_out/r04_fuzz.txt

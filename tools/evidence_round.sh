# round-5 evidence on the final commit (GPU box): everything lands in gpurun_out/, the summaries are copied to profiles/ afterwards
# usage: tools/evidence_round.sh [quick]     quick = suites + bench + kernel stats only
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r05_gputests.log 2>&1; tail -3 gpurun_out/r05_gputests.log
D2G_BS_SPARSE_MIN_N=1 D2G_SP_TILE_FRAC=1 python -m pytest tests -q -m gpu > gpurun_out/r05_gputests_sparse_forced.log 2>&1; tail -3 gpurun_out/r05_gputests_sparse_forced.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; tail -1 gpurun_out/r05_smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench.out 2> gpurun_out/r05_bench.err; tail -1 gpurun_out/r05_bench.out > gpurun_out/r05_bench.json
tools/kstats.sh r05_bench python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-config4 --no-matrices --no-cpu-baseline --no-sketch --no-multiset --no-traffic > /dev/null 2>&1
for m in stated unrelated paired skewed; do for sp in 1 0; do echo -n "N=10000 $m sparse=$sp: "; D2G_BS_SPARSE=$sp MATRIX=$m python tools/k2_time.py 2>/dev/null | grep step; done; done > gpurun_out/r05_k2_matrices.txt
for c in 1 3 10; do echo -n "N=10000 stated + $c collisions per sketch: "; MATRIX=noise C=$c python tools/k2_time.py 2>/dev/null | grep step; done >> gpurun_out/r05_k2_matrices.txt
for sp in 1 0; do echo -n "N=50000 stated sparse=$sp: "; N=50000 D2G_BS_SPARSE=$sp MATRIX=stated python tools/k2_time.py 2>/dev/null | grep step; done >> gpurun_out/r05_k2_matrices.txt
echo -n "N=50000 stated + 1 collision per sketch: " >> gpurun_out/r05_k2_matrices.txt; N=50000 MATRIX=noise C=1 python tools/k2_time.py 2>/dev/null | grep step >> gpurun_out/r05_k2_matrices.txt
N=50000 MATRIX=stated tools/kstats.sh r05_k2_config4 python $GRAFT_REPO_ROOT/tools/k2_time.py > /dev/null 2>&1
[ "$1" = quick ] && exit 0
tools/kstats.sh r05_bench_all_legs python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > /dev/null 2>&1
D2G_BS_SPARSE=0 tools/kstats.sh r05_bench_dense_walk python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-config4 --no-matrices --no-cpu-baseline --no-sketch --no-multiset --no-traffic > /dev/null 2>&1
tools/pmc_round.sh $GRAFT_REPO_ROOT/gpurun_out/r05_pmc.json > gpurun_out/pmc_round.log 2>&1; tail -3 gpurun_out/pmc_round.log
( time python bench.py --gpus 8 --loopback --steps 5 --warmup 2 ) > gpurun_out/r05_bench_w8_loopback.json 2> gpurun_out/r05_bench_w8_loopback.err
( time python bench.py --gpus 2 --loopback --steps 5 --warmup 2 ) > gpurun_out/r05_bench_w2_loopback.json 2> gpurun_out/r05_bench_w2_loopback.err
tools/mgpu_model.sh > gpurun_out/mm.log 2>&1; tail -5 gpurun_out/mm.log
python tools/plist_stats.py > gpurun_out/r05_plist_stats.txt 2>&1; C=1 python tools/plist_stats.py >> gpurun_out/r05_plist_stats.txt 2>&1
D2G_VERBOSE_EXIT=1 python tools/e2e_cli.py --genomes 1000 --threads 112 --big-sketches 50000 > gpurun_out/r05_e2e_cli.txt 2>&1
python tools/fuzz_parity.py 600 515 > gpurun_out/r05_fuzz.txt 2>&1; tail -2 gpurun_out/r05_fuzz.txt
D2G_FUZZ_ONLY=k2,mgpu python tools/fuzz_parity.py 900 616 > gpurun_out/r05_fuzz_k2_long.txt 2>&1; tail -2 gpurun_out/r05_fuzz_k2_long.txt

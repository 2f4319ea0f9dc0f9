# round-6 evidence on the final commit (GPU box): everything lands in gpurun_out/, the summaries are copied to profiles/ afterwards
# usage: tools/evidence_round.sh [quick|rest]     quick = suites + bench + kernel stats + matrices; rest = counters, loopback lines, e2e CLI, fuzz
set -x
MODE=$1          # (the loops below use `set --`)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
if [ "$MODE" != rest ]; then
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06_gputests.log 2>&1; tail -3 gpurun_out/r06_gputests.log
D2G_BS_SPARSE_MIN_N=1 D2G_SP_TILE_FRAC=1 timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06_gputests_sparse_forced.log 2>&1; tail -3 gpurun_out/r06_gputests_sparse_forced.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -1 gpurun_out/r06_smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench.out 2> gpurun_out/r06_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/r06_bench.out > gpurun_out/r06_bench.json
timeout 600 tools/kstats.sh r06_bench python $R/bench.py --steps 20 --warmup 5 --no-config4 --no-matrices --no-cpu-baseline --no-sketch --no-multiset --no-traffic > /dev/null 2>&1
for m in stated unrelated paired skewed; do for sp in 1 0; do echo -n "N=10000 $m sparse=$sp: "; D2G_BS_SPARSE=$sp MATRIX=$m timeout 200 python tools/k2_time.py 2>/dev/null | grep step; done; done > gpurun_out/r06_k2_matrices.txt
for c in 1 3 10 30; do echo -n "N=10000 stated + $c collisions per sketch: "; MATRIX=noise C=$c timeout 200 python tools/k2_time.py 2>/dev/null | grep step; done >> gpurun_out/r06_k2_matrices.txt
for c in 1 3 10; do echo -n "N=10000 stated + $c collisions per sketch, list entry by entry (D2G_SP_LIST_FORM=1): "; D2G_SP_LIST_FORM=1 MATRIX=noise C=$c timeout 200 python tools/k2_time.py 2>/dev/null | grep step; done >> gpurun_out/r06_k2_matrices.txt
for sp in 1 0; do echo -n "N=50000 stated sparse=$sp: "; N=50000 D2G_BS_SPARSE=$sp MATRIX=stated timeout 300 python tools/k2_time.py 2>/dev/null | grep step; done >> gpurun_out/r06_k2_matrices.txt
for c in 1 10; do echo -n "N=50000 stated + $c collisions per sketch: "; N=50000 MATRIX=noise C=$c STEPS=5 timeout 300 python tools/k2_time.py 2>/dev/null | grep step; done >> gpurun_out/r06_k2_matrices.txt
for m in "noise 1" "noise 10" "noise 100" "paired -" "skewed -" "stated -"; do set -- $m; for sp in 1 0; do D2G_BS_SPARSE=$sp MATRIX=$1 C=$2 timeout 120 python3 tools/k2_first.py 2>&1 | grep "first step"; done; done > gpurun_out/r06_k2_first_step.txt
for c in 0 10; do if [ $c = 0 ]; then M=stated; else M=noise; fi; MATRIX=$M C=$c STEPS=20 timeout 300 tools/kstats.sh r06_k2_c$c python3 $R/tools/k2_time.py > /dev/null 2>&1; done
N=50000 MATRIX=stated timeout 600 tools/kstats.sh r06_k2_config4 python $R/tools/k2_time.py > /dev/null 2>&1
for n in 10000 50000; do for m in stated unrelated; do N=$n MATRIX=$m D2G_LIB=$R/dashing2_amd/libd2g_ranktrace.so timeout 200 python3 tools/rank_trace.py 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/r06_rank_trace_final.txt
fi
[ "$MODE" = quick ] && exit 0
timeout 900 tools/kstats.sh r06_bench_all_legs python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > /dev/null 2>&1
D2G_BS_SPARSE=0 timeout 600 tools/kstats.sh r06_bench_dense_walk python $R/bench.py --steps 20 --warmup 5 --no-config4 --no-matrices --no-cpu-baseline --no-sketch --no-multiset --no-traffic > /dev/null 2>&1
timeout 1200 tools/pmc_round.sh $R/gpurun_out/r06_pmc.json > gpurun_out/pmc_round.log 2>&1; tail -3 gpurun_out/pmc_round.log
( time timeout 900 python bench.py --gpus 8 --loopback --steps 5 --warmup 2 ) > gpurun_out/r06_bench_w8_loopback.json 2> gpurun_out/r06_bench_w8_loopback.err
( time timeout 900 python bench.py --gpus 2 --loopback --steps 5 --warmup 2 ) > gpurun_out/r06_bench_w2_loopback.json 2> gpurun_out/r06_bench_w2_loopback.err
timeout 300 python tools/plist_stats.py > gpurun_out/r06_plist_stats.txt 2>&1; C=1 timeout 300 python tools/plist_stats.py >> gpurun_out/r06_plist_stats.txt 2>&1
D2G_VERBOSE_EXIT=1 timeout 900 python tools/e2e_cli.py --genomes 1000 --threads 112 --big-sketches 50000 > gpurun_out/r06_e2e_cli.txt 2>&1
timeout 700 python tools/fuzz_parity.py 600 616 > gpurun_out/r06_fuzz.txt 2>&1; tail -2 gpurun_out/r06_fuzz.txt
D2G_FUZZ_ONLY=k2,mgpu timeout 1000 python tools/fuzz_parity.py 900 717 > gpurun_out/r06_fuzz_k2_long.txt 2>&1; tail -2 gpurun_out/r06_fuzz_k2_long.txt

set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/r03_gputests.log 2>&1; tail -3 gpurun_out/r03_gputests.log
tools/pmc_round.sh > gpurun_out/pmc_round.log 2>&1; tail -3 gpurun_out/pmc_round.log
cp gpurun_out/r03_pmc.json profiles/r03_pmc.json
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench.out 2> gpurun_out/r03_bench.err; tail -1 gpurun_out/r03_bench.out > gpurun_out/r03_bench.json
tools/kstats.sh r03_bench python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-config4 --no-matrices --no-cpu-baseline --no-sketch --no-multiset > /dev/null 2>&1
tools/kstats.sh r03_bench_all_legs python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
tools/mgpu_model.sh > gpurun_out/mm.log 2>&1; tail -5 gpurun_out/mm.log
D2G_VERBOSE_EXIT=1 python tools/e2e_cli.py --genomes 1000 --threads 112 --big-sketches 50000 > gpurun_out/r03_e2e_cli.txt 2>&1
(echo; echo "---- the same with the device parser (D2G_DEVICE_PARSE=1)"; D2G_DEVICE_PARSE=1 D2G_VERBOSE_EXIT=1 python tools/e2e_cli.py --genomes 1000 --threads 112 --sketches 10 2>&1 | head -16) >> gpurun_out/r03_e2e_cli.txt
python tools/k0_time.py 200 5 > gpurun_out/r03_k0_time.txt 2>&1
python tools/fuzz_parity.py 600 303 > gpurun_out/r03_fuzz.txt 2>&1; tail -2 gpurun_out/r03_fuzz.txt

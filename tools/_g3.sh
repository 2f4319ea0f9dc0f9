cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_cli.py -x -q -k "multi_gpu or gpu_stats or parallel_write or batches" 2>&1 | tail -25
D2G_VERBOSE_EXIT=1 python tools/e2e_cli.py --genomes 1000 --threads 112 --big-sketches 50000 > gpurun_out/r04/e2e_cli_a.txt 2>&1; grep -n "cmp\|sketch run\|PHYLIP run" gpurun_out/r04/e2e_cli_a.txt | grep -v "\[d2g\]" | head -30

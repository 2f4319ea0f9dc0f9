bash tools/evidence_round.sh quick 2>&1 | grep -v "^+" | tail -30
head -20 gpurun_out/r05_bench_kernel_stats.txt; cat gpurun_out/r05_k2_matrices.txt | cut -c1-120

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/fuzz_parity.py 1800 3031 > gpurun_out/r05_fuzz_all_long.txt 2>&1; tail -2 gpurun_out/r05_fuzz_all_long.txt

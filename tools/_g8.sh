cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
D2G_FUZZ_ONLY=k2 python tools/fuzz_parity.py 150 404 2>&1 | tail -5

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c21
D2G_FUZZ_ONLY=k2,mgpu timeout 700 python tools/fuzz_parity.py 420 717 > gpurun_out/c21/fuzz_k2.txt 2>&1; tail -3 gpurun_out/c21/fuzz_k2.txt

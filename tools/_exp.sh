cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x > gpurun_out/exp3_gputests.log 2>&1; tail -4 gpurun_out/exp3_gputests.log
D2G_FUZZ_ONLY=k2 python tools/fuzz_parity.py 90 7171 > gpurun_out/exp3_fuzz_k2.txt 2>&1; tail -3 gpurun_out/exp3_fuzz_k2.txt

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for v in A B C D A B; do python tools/cmp_setup_time2.py $v; done > gpurun_out/r04/cmp_setup_time2.txt 2>&1; cat gpurun_out/r04/cmp_setup_time2.txt

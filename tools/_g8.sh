cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/k3_refine_proto.hip -o /tmp/k3_refine_proto && /tmp/k3_refine_proto > gpurun_out/r04/k3_refine_proto.txt 2>&1; cat gpurun_out/r04/k3_refine_proto.txt
python tools/k3_time.py 250 5000000 21 2048 3 2>&1 | tail -6 > gpurun_out/r04/k3_time_default.txt; cat gpurun_out/r04/k3_time_default.txt
D2G_K3_COMPACT=1 python tools/k3_time.py 250 5000000 21 2048 3 2>&1 | tail -6 > gpurun_out/r04/k3_time_compact.txt; cat gpurun_out/r04/k3_time_compact.txt
D2G_K3_COMPACT=1 tools/kstats.sh r04_k3c python $GRAFT_REPO_ROOT/tools/k3_time.py 250 5000000 21 2048 3 > /dev/null 2>&1; head -14 gpurun_out/r04_k3c_kernel_stats.txt | cut -c1-50,92-150

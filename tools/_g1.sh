set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_mgpu.py -x -q 2>&1 | tail -8 > gpurun_out/r04/t_mgpu.log; cat gpurun_out/r04/t_mgpu.log
( time timeout 600 python bench.py --gpus 8 --loopback --steps 5 --warmup 2 ) > gpurun_out/r04/bench_w8_loop.json 2> gpurun_out/r04/bench_w8_loop.err; tail -3 gpurun_out/r04/bench_w8_loop.err
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r04/bench_n1.json 2> gpurun_out/r04/bench_n1.err; tail -5 gpurun_out/r04/bench_n1.err
( time timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 ) > gpurun_out/r04/bench_n2_refuse.json 2>&1

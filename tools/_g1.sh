timeout 1500 python -m pytest tests/test_gpu_k2.py tests/test_gpu_mgpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
D2G_BS_SPARSE_MIN_N=1 D2G_SP_SEG_DIV=1 timeout 1500 python -m pytest tests/test_gpu_k2.py tests/test_gpu_mgpu.py tests/test_gpu_cli.py -x -q -m gpu -k "not config4" 2>&1 | grep -E "passed|failed"
D2G_FUZZ_ONLY=k2,mgpu timeout 400 python tools/fuzz_parity.py 200 99 2>&1 | tail -1
bash tools/mgpu_model.sh 2>&1 | grep "^W=8\|^W=1" | cut -c1-330
grep "sp_rows" gpurun_out/r04_mgpu_model.txt | tail -1 | cut -c1-300
python tools/k2_time.py 2>&1 | tail -2 | head -1 | cut -c1-60

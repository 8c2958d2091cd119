mkdir -p gpurun_out/r05
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_golden.py tests/test_gpu_fullscale.py tests/test_gpu_scale_configs.py tests/test_gpu_comm.py -m gpu -x -q > gpurun_out/r05/pytest_part.log 2>&1; grep -E "passed|failed" gpurun_out/r05/pytest_part.log
bash tools/r05_ab2.sh | cut -c1-170

python3 -m pytest tests/test_baseline_gpu.py -m gpu -q -x -k "band_kernels" 2>&1 | grep -E "^E  |passed|failed|HIP fp32" | head -12 > gpurun_out/r05x2.txt
python3 -m pytest tests/test_losses_gpu.py tests/test_main_gpu.py tests/test_metrics_gpu.py tests/test_mmnist_gpu.py tests/test_optim_gpu.py tests/test_safety_gpu.py tests/test_step_gpu.py -m gpu -q 2>&1 | tail -8 >> gpurun_out/r05x2.txt

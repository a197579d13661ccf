mkdir -p gpurun_out/r05
python3 -m pytest tests/ -m gpu -q 2>&1 | tail -6 > gpurun_out/r05/r05_gpu_tests.txt

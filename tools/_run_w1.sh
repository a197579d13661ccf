python3 -m pytest tests/test_baseline_gpu.py -m gpu -q -s -k "through_the_16bit" 2>&1 | grep -E "^E  |passed|failed|HIP fp32" | head -12 > gpurun_out/r05w20.txt

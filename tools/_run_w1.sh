for i in 1 2 3 4 5; do
python3 -m pytest tests/test_losses_gpu.py -m gpu -q -x -k recorded_waveeq 2>&1 | grep -E "^E   +Assert|passed|failed" | head -3
done > gpurun_out/r05x6.txt

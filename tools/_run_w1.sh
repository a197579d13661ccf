python3 -m pytest tests/test_ddp_gpu.py -m gpu -q -x -k sharded 2>&1 | grep -E "^E |Error|assert|passed|failed" | head -20 > gpurun_out/r05w6.txt

export VARSEP_BENCH_LIVE_PROFILE=0
out=gpurun_out/r05w18.txt
: > $out
python3 -m pytest tests/test_baseline_gpu.py -m gpu -q -s -k "through_the_16bit" 2>&1 | grep -E "^E  |passed|failed|HIP fp32" | head -12 >> $out
python3 -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "tap or convt or dcgan" 2>&1 | tail -2 >> $out
b() { python3 bench.py --config $2 --extra_configs none --no_cpu_baseline --steps 20 2>>gpurun_out/r05w18.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', d['ms_per_step'], 'ms')" >> $out; }
b "final" mnist_b128
b "final" mnist_b128

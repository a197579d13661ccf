export VARSEP_BENCH_LIVE_PROFILE=0
python3 -m pytest tests/test_safety_gpu.py tests/test_gemm_gpu.py -m gpu -x -q 2>&1 | tail -15
for cfg in mnist_b128; do
for v in 0 1; do
VS_BAND_V2=$v python3 bench.py --config $cfg --extra_configs none --no_cpu_baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg VS_BAND_V2=$v', d['ms_per_step'], 'ms')"
done
done
bash tools/_run_wave.sh

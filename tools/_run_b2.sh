export VARSEP_BENCH_LIVE_PROFILE=0
python3 -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "img16 or res_block or conv3" 2>&1 | tail -3
for i in 1 2; do
for v in 0 1; do
VS_IMG16_WIDE_STORE=$v python3 bench.py --config sst --extra_configs none --no_cpu_baseline --steps 6 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sst VS_IMG16_WIDE_STORE=$v', d['ms_per_step'], 'ms')"
done
done

export VS_BAND_BENCH_VARIANTS='VS_WGRAD_V2=0,1'
python3 tools/band_bench.py wgrad --check "--only=192->128,260->256,384->128,64->64 @32 dec,512->512 @4 dec" 2>&1 | grep -v amdgpu.ids
export VARSEP_BENCH_LIVE_PROFILE=0
for cfg in taxibj sst mnist_b128; do
for v in 0 1; do
VS_WGRAD_V2=$v python3 bench.py --config $cfg --extra_configs none --no_cpu_baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg VS_WGRAD_V2=$v', d['ms_per_step'], 'ms')"
done
done

export VARSEP_BENCH_LIVE_PROFILE=0
out=gpurun_out/r05w12.txt
: > $out
python3 -m pytest tests/test_ddp_gpu.py -m gpu -q -x -k "conv_family or segments" 2>&1 | grep -E "^E  |passed|failed" | head -8 >> $out
b() { python3 bench.py --config $2 --extra_configs none --no_cpu_baseline --steps 20 2>>gpurun_out/r05w12.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', d['ms_per_step'], 'ms')" >> $out; }
for cfg in taxibj sst; do
VARSEP_BENCH_FORCE_DIST=1 VARSEP_BATCH_SMALL_ADDS=0 b "dist1 adds per block" $cfg
VARSEP_BENCH_FORCE_DIST=1 b "dist1 adds batched  " $cfg
done

export VARSEP_BENCH_LIVE_PROFILE=0
out=gpurun_out/r05w15.txt
: > $out
b() { python3 bench.py --config waveeq --extra_configs none --no_cpu_baseline --steps 20 2>>gpurun_out/r05w15.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms')" >> $out; }
for i in 1 2; do
b "default queues"
GPU_MAX_HW_QUEUES=2 b "queues 2"
GPU_MAX_HW_QUEUES=3 b "queues 3"
GPU_MAX_HW_QUEUES=4 b "queues 4"
GPU_MAX_HW_QUEUES=6 b "queues 6"
GPU_MAX_HW_QUEUES=8 b "queues 8"
VARSEP_WGRAD_LANES=2 b "lanes 2"
done

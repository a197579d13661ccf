export VARSEP_BENCH_LIVE_PROFILE=0
out=gpurun_out/r05w14.txt
: > $out
b() { python3 bench.py --config waveeq --extra_configs none --no_cpu_baseline --steps 20 2>>gpurun_out/r05w14.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms')" >> $out; }
for i in 1 2 3 4; do
VARSEP_ROLLOUT_NODE_LAST=0 b "rollout node first (old)"
VARSEP_ROLLOUT_NODE_LAST=1 b "rollout node last  (new)"
done
VARSEP_ROLLOUT_NODE_LAST=1 bash tools/_prof_one.sh w14new waveeq

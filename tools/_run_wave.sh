export VARSEP_BENCH_LIVE_PROFILE=0
for i in 1 2 3; do
for s in 0 1; do
VARSEP_TAIL_SPLIT=$s python3 bench.py --extra_configs none --no_cpu_baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TAIL_SPLIT=$s', d['ms_per_step'], d['repeats'] if 'repeats' in d else '')"
done
done

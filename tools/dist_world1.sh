#!/bin/bash
# The N > 1 code path (process group, flat buckets, RCCL collectives per step, sharded optimizer for the MLP family) forced at world size 1 beside
# the plain single-process step, interleaved on one box: what the data-parallel machinery costs before a byte crosses xGMI.
# usage: bash tools/dist_world1.sh [workloads...] > gpurun_out/rNN_dist_world1.txt      (default: waveeq taxibj sst)
cd "$(dirname "$0")/.." || exit 2
wl=${@:-waveeq taxibj sst}
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', '$2', d['ms_per_step'], 'ms', d['config']['grad_allreduce'])"; }
for w in $wl; do
  for r in 1 2; do
    python3 bench.py --config $w --extra_configs none --no_cpu_baseline --steps 10 2>/dev/null | line plain $w
    VARSEP_BENCH_FORCE_DIST=1 python3 bench.py --config $w --extra_configs none --no_cpu_baseline --steps 10 2>/dev/null | line "N>1 path at world size 1 (RCCL)" $w
  done
done

export TMPDIR=/tmp
mkdir -p gpurun_out/r05
bash tools/collect_profiles.sh r05 waveeq mnist_b128 taxibj sst sst_fp16 > gpurun_out/r05_collect.log 2>&1
out=gpurun_out/r05/r05_dist_world1.txt
: > $out
export VARSEP_BENCH_LIVE_PROFILE=0
b() { python3 bench.py --config $2 --extra_configs none --no_cpu_baseline --steps 20 2>>gpurun_out/r05/dist.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', d['ms_per_step'], 'ms', d['config'].get('grad_allreduce'))" >> $out; }
for cfg in waveeq taxibj sst; do
b "plain" $cfg
VARSEP_BENCH_FORCE_DIST=1 b "N>1 path at world size 1 (RCCL)" $cfg
b "plain" $cfg
VARSEP_BENCH_FORCE_DIST=1 b "N>1 path at world size 1 (RCCL)" $cfg
done
VARSEP_BENCH_FORCE_DIST=1 VARSEP_SHARD_OPT=0 b "N>1 path at world size 1, replicated update (VARSEP_SHARD_OPT=0)" waveeq
unset VARSEP_BENCH_LIVE_PROFILE
bash tools/collect_mfma_util.sh r05 taxibj mnist_b128 sst waveeq > gpurun_out/r05_util.log 2>&1
python3 tools/band_bench.py all > gpurun_out/r05/r05_band_bench.txt 2>gpurun_out/r05/band.err
VARSEP_BENCH_SHARE_GPU=1 VARSEP_BENCH_LIVE_PROFILE=0 timeout 900 python3 bench.py --gpus 2 --no_cpu_baseline > gpurun_out/r05/r05_two_ranks_one_gpu_gloo_bench.json 2> gpurun_out/r05/two_ranks.err
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/r05_smoke.txt 2>&1
python3 -m pytest tests/ -m gpu -q 2>&1 | tail -6 > gpurun_out/r05/r05_gpu_tests.txt

export TMPDIR=/tmp
bash tools/collect_mfma_util.sh r05 taxibj mnist_b128 sst waveeq > gpurun_out/r05_util.log 2>&1
python3 tools/band_bench.py all > gpurun_out/r05/r05_band_bench.txt 2>gpurun_out/r05/band.err
VARSEP_BENCH_SHARE_GPU=1 VARSEP_BENCH_LIVE_PROFILE=0 timeout 900 python3 bench.py --gpus 2 --no_cpu_baseline > gpurun_out/r05/r05_two_ranks_one_gpu_gloo_bench.json 2> gpurun_out/r05/two_ranks.err
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/r05_smoke.txt 2>&1
python3 -m pytest tests/ -m gpu -q 2>&1 | tail -12 > gpurun_out/r05/r05_gpu_tests.txt

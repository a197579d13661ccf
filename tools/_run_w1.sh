mkdir -p gpurun_out/r05
python3 bench.py > gpurun_out/r05/r05_bench_default.json 2> gpurun_out/r05/bench_default.err

export VARSEP_BENCH_LIVE_PROFILE=0
out=gpurun_out/r05w17.txt
: > $out
python3 -m pytest tests/test_gemm_gpu.py -m gpu -q -x 2>&1 | tail -2 >> $out
b() { python3 bench.py --config $2 --extra_configs none --no_cpu_baseline --steps 20 2>>gpurun_out/r05w17.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', d['ms_per_step'], 'ms')" >> $out; }
for i in 1 2 3; do
VS_GEMM_DB=0 b "single buffer" waveeq
VS_GEMM_DB=1 b "double buffer" waveeq
done
VS_GEMM_DB=0 b "single buffer" mnist_b128
VS_GEMM_DB=1 b "double buffer" mnist_b128
VS_GEMM_DB=0 b "single buffer" taxibj
VS_GEMM_DB=1 b "double buffer" taxibj
for v in 0 1; do echo "VS_GEMM_DB=$v" >> $out; VS_GEMM_DB=$v python3 tools/gemm_bench.py bf16 cold 2>&1 | grep -E "dec fwd 1200->1200|dec dgrad 1200->1200|dec dgrad 4096|enc fwd 1200->1200|dec fwd 32" | cut -c1-110 >> $out; done

export VARSEP_BENCH_LIVE_PROFILE=0
out=gpurun_out/r05w9.txt
: > $out
b() { python3 bench.py --config waveeq --extra_configs none --no_cpu_baseline --steps 20 2>>gpurun_out/r05w9.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms')" >> $out; }
for i in 1 2 3; do
b "base          "
VS_GEMM_T64_BELOW_RR=400 b "fwd 128x64    "
VS_GEMM_T64_BELOW=400 b "all 128x64    "
VS_GEMM_MID=0 b "no mid        "
done

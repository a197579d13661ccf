export VARSEP_BENCH_LIVE_PROFILE=0
out=gpurun_out/r05w16.txt
: > $out
b() { python3 bench.py --config waveeq --extra_configs none --no_cpu_baseline --steps 20 2>>gpurun_out/r05w16.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms')" >> $out; }
for i in 1 2 3; do
b "base                    "
VS_GEMM_MID_SPLIT2=1 VS_GEMM_SPLITK_FUSED=1 b "mid split2 + fused <=128K"
VS_GEMM_MID_SPLIT2=1 b "mid split2 + reduce launch"
done
VS_GEMM_MID_SPLIT2=1 VS_GEMM_SPLITK_FUSED=1 bash tools/_prof_one.sh w16new waveeq

export VARSEP_BENCH_LIVE_PROFILE=0
out=gpurun_out/r05w3.txt
: > $out
python3 -m pytest tests/test_gemm_gpu.py -m gpu -q -x -k "splitk or adam" 2>&1 | tail -3 >> $out
b() { python3 bench.py --config waveeq --extra_configs none --no_cpu_baseline --steps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms')" >> $out; }
for i in 1 2 3; do
VS_ADAM_PIPE=0 VS_GEMM_SPLITK_FUSED=0 b "base      "
VS_ADAM_PIPE=1 VS_GEMM_SPLITK_FUSED=0 b "pipe      "
VS_ADAM_PIPE=0 VS_GEMM_SPLITK_FUSED=1 b "fused<=96K"
VS_ADAM_PIPE=1 VS_GEMM_SPLITK_FUSED=1 b "pipe+fused"
done
VS_ADAM_PIPE=1 VS_GEMM_SPLITK_FUSED=1 bash tools/_prof_one.sh w3new waveeq
python3 tools/host_vs_gpu.py waveeq >> $out 2>&1

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/a_tests.txt
for v in 1 0 1 0; do
  echo "VS_GEMM_P8=$v" >> gpurun_out/a_bench.txt
  VS_GEMM_P8=$v python bench.py --config waveeq --extra_configs none --no_cpu_baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['us_per_step'])" >> gpurun_out/a_bench.txt 2>&1
done
echo "NI1" >> gpurun_out/a_bench.txt
for v in 1 0; do
VS_GEMM_P8_NI=$v python bench.py --config waveeq --extra_configs none --no_cpu_baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['us_per_step'])" >> gpurun_out/a_bench.txt 2>&1
done
python tools/gemm_bench.py bf16 cold > gpurun_out/a_gemm_cold.txt 2>&1
cat gpurun_out/a_tests.txt gpurun_out/a_bench.txt gpurun_out/a_gemm_cold.txt

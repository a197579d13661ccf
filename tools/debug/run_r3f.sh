mkdir -p gpurun_out/r3f
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r3f/all_gpu_tests.log 2>&1
tail -5 gpurun_out/r3f/all_gpu_tests.log
B="python bench.py --extra_configs none --no_cpu_baseline"
for w in mnist_b128 taxibj sst; do $B --config $w > gpurun_out/r3f/bench_$w.json 2>/dev/null; done
for w in mnist_b128 taxibj; do
  rm -rf gpurun_out/r3f/p
  VARSEP_BENCH_NO_EVENTS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3f/p -o p -- python3 bench.py --config $w --no_cpu_baseline --extra_configs none --steps 20 --repeats 2 > gpurun_out/r3f/prof_$w.log 2>&1
  f=$(find gpurun_out/r3f/p -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/r3f/${w}_kernel_stats.csv
  python3 tools/prof_summary.py $f > gpurun_out/r3f/${w}_summary.md
  python3 tools/replay_stats.py $f gpurun_out/r3f/${w}_replay.json > gpurun_out/r3f/${w}_replay.txt
done
rm -rf gpurun_out/r3f/p
for f in gpurun_out/r3f/bench_*.json; do python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step_all'])" $f; done
head -14 gpurun_out/r3f/mnist_b128_replay.txt

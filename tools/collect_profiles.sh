#!/bin/bash
# Regenerates the per-round evidence under gpurun_out/<tag>/ on a GPU box (copy what is to be judged into profiles/):
#   bench lines (un-profiled), rocprofv3 kernel statistics + per-step summaries of the four workloads, the WaveEq step timeline,
#   the PMC FETCH_SIZE / WRITE_SIZE passes (eager steps: counters are per dispatch) and the traffic tables.
# usage: bash tools/collect_profiles.sh r02
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/${tag}_bench_default.json 2> $out/bench_default.err
for w in waveeq mnist_b128 taxibj sst; do
  python3 bench.py --config $w --no_cpu_baseline --extra_configs none > $out/${tag}_${w}_bf16_bench.json 2>/dev/null
done
python3 bench.py --precision fp16 --no_cpu_baseline --extra_configs none > $out/${tag}_waveeq_fp16_bench.json 2>/dev/null
python3 bench.py --precision fp32 --no_cpu_baseline --extra_configs none > $out/${tag}_waveeq_fp32_bench.json 2>/dev/null
VARSEP_BENCH_FORCE_DIST=1 python3 bench.py --no_cpu_baseline --extra_configs none > $out/${tag}_waveeq_bf16_dist_world1_bench.json 2>/dev/null
for w in waveeq mnist_b128 taxibj sst; do
  rm -rf $out/p
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o p -- python3 bench.py --config $w --no_cpu_baseline --extra_configs none --steps 20 --repeats 2 > $out/prof_$w.log 2>&1
  f=$(find $out/p -name "*kernel_stats.csv" | head -1)
  cp $f $out/${tag}_${w}_bf16_kernel_stats.csv
  python3 tools/prof_summary.py $f > $out/${tag}_${w}_bf16_summary.md
  if [ $w = waveeq ]; then
    t=$(find $out/p -name "*kernel_trace.csv" | head -1)
    python3 tools/step_timeline.py $t 20 > $out/${tag}_waveeq_bf16_timeline.txt
  fi
done
rm -rf $out/p
# (every profiler call is wrapped in `timeout`: once the eager SST step aborted under --pmc with HSA_STATUS_ERROR_INVALID_PACKET_FORMAT and
#  rocprofv3 hung while finalizing.  The PMC dispatch path rejects the 144 KiB-LDS launch of the default few-maps kernel, so the SST passes
#  run with VS_CONV_IMG_PAIR=1: 72 KiB per workgroup, twice the slabs)
for w in waveeq taxibj mnist_b128 sst; do
  rm -rf $out/f $out/w
  if [ $w = sst ]; then export VS_CONV_IMG_PAIR=1; fi
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -o p -- python3 bench.py --config $w --no_graph --steps 6 --warmup 2 --repeats 1 --no_cpu_baseline --extra_configs none > $out/pmc_f_$w.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -o p -- python3 bench.py --config $w --no_graph --steps 6 --warmup 2 --repeats 1 --no_cpu_baseline --extra_configs none > $out/pmc_w_$w.log 2>&1
  python3 tools/pmc_traffic.py $(find $out/f -name "*counter_collection.csv" | head -1) $(find $out/w -name "*counter_collection.csv" | head -1) $out/${tag}_${w}_bf16_traffic.json $out/${tag}_${w}_bf16_hbm_traffic.md $w
  unset VS_CONV_IMG_PAIR
done
rm -rf $out/f $out/w
ls -la $out

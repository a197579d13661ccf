#!/bin/bash
# Regenerates the per-round evidence under gpurun_out/<tag>/ on a GPU box (copy what is to be judged into profiles/):
#   bench lines (un-profiled), rocprofv3 kernel statistics of the REPLAYED step of every workload (+ per-group replay tables that bench.py
#   attaches to its roofline objects, + per-step summaries), the WaveEq step timeline, the PMC FETCH_SIZE / WRITE_SIZE passes (eager
#   steps: counters are per dispatch) and the traffic tables.
# usage: bash tools/collect_profiles.sh r03 [workloads...]        (default workloads: waveeq mnist_b128 taxibj sst sst_fp16)
tag=${1:-r03}
shift
wl=${@:-waveeq mnist_b128 taxibj sst sst_fp16}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for w in $wl; do
  cfg=${w%_fp16}; prec=bf16; if [ $cfg != $w ]; then prec=fp16; fi
  name=${tag}_${cfg}_${prec}
  rm -rf $out/p
  # VARSEP_BENCH_NO_EVENTS=1: no instrumented eager steps, so the statistics are those of the replayed step (plus the recording's warm-up)
  VARSEP_BENCH_NO_EVENTS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o p -- python3 bench.py --config $cfg --precision $prec --no_cpu_baseline --extra_configs none --steps 20 --repeats 2 > $out/prof_$w.log 2>&1
  f=$(find $out/p -name "*kernel_stats.csv" | head -1)
  cp $f $out/${name}_kernel_stats.csv
  python3 tools/prof_summary.py $f > $out/${name}_summary.md
  python3 tools/replay_stats.py $f $out/${name}_replay.json ${name}_kernel_stats.csv > $out/${name}_replay.txt
  if [ $w = waveeq ]; then
    t=$(find $out/p -name "*kernel_trace.csv" | head -1)
    python3 tools/step_timeline.py $t 20 > $out/${name}_timeline.txt
  fi
done
rm -rf $out/p
# (every profiler call is wrapped in `timeout`: once the eager SST step aborted under --pmc with HSA_STATUS_ERROR_INVALID_PACKET_FORMAT and
#  rocprofv3 hung while finalizing.  The PMC dispatch path rejects the 144 KiB-LDS launch of the default few-maps kernel, so the SST passes
#  run with VS_CONV_IMG_PAIR=1: 72 KiB per workgroup, twice the slabs)
for w in $wl; do
  cfg=${w%_fp16}; prec=bf16; if [ $cfg != $w ]; then prec=fp16; fi
  name=${tag}_${cfg}_${prec}
  rm -rf $out/f $out/w
  if [ $cfg = sst ]; then export VS_CONV_IMG_PAIR=1; fi
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -o p -- python3 bench.py --config $cfg --precision $prec --no_graph --steps 6 --warmup 2 --repeats 1 --no_cpu_baseline --extra_configs none > $out/pmc_f_$w.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -o p -- python3 bench.py --config $cfg --precision $prec --no_graph --steps 6 --warmup 2 --repeats 1 --no_cpu_baseline --extra_configs none > $out/pmc_w_$w.log 2>&1
  python3 tools/pmc_traffic.py $(find $out/f -name "*counter_collection.csv" | head -1) $(find $out/w -name "*counter_collection.csv" | head -1) $out/${name}_traffic.json $out/${name}_hbm_traffic.md $w
  unset VS_CONV_IMG_PAIR
done
rm -rf $out/f $out/w
# bench lines last: they pick up the replay / traffic tables once those are copied into profiles/ (run again after copying)
python3 bench.py > $out/${tag}_bench_default.json 2> $out/bench_default.err
ls -la $out

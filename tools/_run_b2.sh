tag=r05
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for w in taxibj sst mnist_b128 waveeq; do
  cfg=$w; prec=bf16
  name=${tag}_${cfg}_${prec}
  rm -rf $out/p
  VARSEP_BENCH_NO_EVENTS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o p -- python3 bench.py --config $cfg --precision $prec --no_cpu_baseline --extra_configs none --steps 20 --repeats 2 > $out/prof_$w.log 2>&1
  f=$(find $out/p -name "*kernel_stats.csv" | head -1)
  cp $f $out/${name}_kernel_stats.csv
  python3 tools/prof_summary.py $f > $out/${name}_summary.md
  python3 tools/replay_stats.py $f $out/${name}_replay.json ${name}_kernel_stats.csv > $out/${name}_replay.txt
done
rm -rf $out/p

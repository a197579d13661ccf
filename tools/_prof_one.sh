# usage: bash tools/_prof_one.sh <tag> <config> [precision]   (environment switches are inherited)
tag=$1; cfg=$2; prec=${3:-bf16}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
rm -rf $out/p
VARSEP_BENCH_NO_EVENTS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o p -- python3 bench.py --config $cfg --precision $prec --no_cpu_baseline --extra_configs none --steps 20 --repeats 2 > $out/prof.log 2>&1
f=$(find $out/p -name "*kernel_stats.csv" | head -1)
cp $f $out/${tag}_kernel_stats.csv
python3 tools/prof_summary.py $f > $out/${tag}_summary.md
python3 tools/replay_stats.py $f $out/${tag}_replay.json ${tag}_kernel_stats.csv > $out/${tag}_replay.txt
t=$(find $out/p -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $t > $out/${tag}_timeline.txt 2>/dev/null
rm -rf $out/p

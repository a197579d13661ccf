cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B=tools/probes/p8_bench
O=gpurun_out/b_p8.txt
: > $O
run() { echo "== $*" >> $O; timeout 300 $B "$@" >> $O 2>&1; echo "rc=$?" >> $O; }
for l in "0 1" "1 1" "1 0" "0 0"; do
run 256 256 64 $l 1 int 3
run 512 768 192 $l 1 int 3
run 1000 1016 1000 $l 1 int 5
run 3328 1200 4096 $l 1 int 5
run 3328 1200 1200 $l 1 cold 20
run 3328 1200 4096 $l 1 cold 20
run 4096 1200 3328 $l 1 cold 20
done
grep -c "check ok" $O; grep -c "MISMATCH\|WRONG" $O; grep "^p8" $O | grep cold
out=gpurun_out/b
mkdir -p $out; rm -rf $out/p
VARSEP_BENCH_NO_EVENTS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -o p -- python3 bench.py --config waveeq --precision bf16 --no_cpu_baseline --extra_configs none --steps 20 --repeats 2 > $out/prof.log 2>&1
t=$(find $out/p -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $t 20 > gpurun_out/b_timeline.txt
f=$(find $out/p -name "*kernel_stats.csv" | head -1)
python3 tools/replay_stats.py $f $out/replay.json x > gpurun_out/b_replay.txt
rm -rf $out/p
python3 tools/host_vs_gpu.py waveeq 200 > gpurun_out/b_host.txt 2>&1
tail -3 gpurun_out/b_host.txt

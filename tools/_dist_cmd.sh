export VARSEP_BENCH_LIVE_PROFILE=0
python bench.py --config taxibj --extra_configs sst,waveeq --no_cpu_baseline > gpurun_out/r4g_plain.json 2>/dev/null
VARSEP_BENCH_FORCE_DIST=1 python bench.py --config taxibj --extra_configs sst,waveeq --no_cpu_baseline > gpurun_out/r4g_dist_seg.json 2> gpurun_out/r4g_dist_seg.err
VARSEP_BENCH_FORCE_DIST=1 VARSEP_GRAPH_SEGMENTS=0 python bench.py --config taxibj --extra_configs sst,waveeq --no_cpu_baseline > gpurun_out/r4g_dist_oneseg.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
VARSEP_BENCH_FORCE_DIST=1 VARSEP_BENCH_NO_EVENTS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config taxibj --no_cpu_baseline --extra_configs none --steps 10 --repeats 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
t=$(find /tmp/pd -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $t 4 > gpurun_out/r4g_taxibj_dist_timeline.txt 2>&1
grep -n -i "rccl\|nccl\|AllReduce\|ncclDev" gpurun_out/r4g_taxibj_dist_timeline.txt | head -5
for f in plain dist_seg dist_oneseg; do python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r4g_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['config']['workload'][:8], d['ms_per_step'], {k:v.get('ms_per_step', v) for k,v in d.get('configs',{}).items()})
" $f; done
tail -3 gpurun_out/r4g_dist_seg.err | cut -c1-300

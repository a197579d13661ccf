cd $GRAFT_REPO_ROOT
O=gpurun_out/c_bench.txt
: > $O
one() { echo "$1" >> $O; env $1 python bench.py --config waveeq --extra_configs none --no_cpu_baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['us_per_step'])" >> $O 2>&1; }
for r in 1 2; do
one "X=0"
one "VS_GEMM_P8_NI=1"
one "VARSEP_FUSE_FRAME_LOSS=1"
one "VS_GEMM_P8_NI=1 VARSEP_FUSE_FRAME_LOSS=1"
done
cat $O

mkdir -p gpurun_out/r3g
python -m pytest tests/test_conv_gpu.py tests/test_step_gpu.py tests/test_ddp_gpu.py tests/test_fp16_gpu.py -x -q > gpurun_out/r3g/tests.log 2>&1
tail -4 gpurun_out/r3g/tests.log
B="python bench.py --extra_configs none --no_cpu_baseline"
for w in mnist_b128 taxibj sst; do $B --config $w > gpurun_out/r3g/bench_$w.json 2>/dev/null; done
VARSEP_PREPACK_CONV=0 $B --config taxibj > gpurun_out/r3g/bench_taxibj_noprepack.json 2>/dev/null
$B > gpurun_out/r3g/wave_base.json 2>/dev/null
VARSEP_FUSE_ADAM_MIN=1000000 $B > gpurun_out/r3g/wave_fuse1m.json 2>/dev/null
VARSEP_FUSE_ADAM_MIN=30000 $B > gpurun_out/r3g/wave_fuse30k.json 2>/dev/null
for f in gpurun_out/r3g/*.json; do python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step_all'], d['config']['final_loss'])" $f; done

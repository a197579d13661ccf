mkdir -p gpurun_out/r3d
python -m pytest tests/test_conv_gpu.py -q -k "k4s2" > gpurun_out/r3d/k4s2_ops.log 2>&1
python -m pytest tests/test_blocks_full_gpu.py tests/test_baseline_gpu.py -q -s -k "mnist_b128 and (lowp or bf16)" > gpurun_out/r3d/mnist_parity.log 2>&1
python -m pytest tests/test_ddp_gpu.py -q -s -k "conv_family" > gpurun_out/r3d/ddp.log 2>&1
B="python bench.py --extra_configs none --no_cpu_baseline"
VS_CONV_K4S2=0 $B --config mnist_b128 > gpurun_out/r3d/mnist_k4s2_off.json 2>/dev/null
$B --config mnist_b128 > gpurun_out/r3d/mnist_k4s2_on.json 2>/dev/null
$B > gpurun_out/r3d/wave_base.json 2>/dev/null
VARSEP_ADAM_EARLY_BUCKET=1 $B > gpurun_out/r3d/wave_early.json 2>/dev/null
VARSEP_FUSED_AFTER_ROLLOUT=1 $B > gpurun_out/r3d/wave_late.json 2>/dev/null
VARSEP_ADAM_EARLY_BUCKET=1 VARSEP_FUSED_AFTER_ROLLOUT=1 $B > gpurun_out/r3d/wave_both.json 2>/dev/null
$B > gpurun_out/r3d/wave_base2.json 2>/dev/null
tail -3 gpurun_out/r3d/k4s2_ops.log; tail -3 gpurun_out/r3d/mnist_parity.log; tail -6 gpurun_out/r3d/ddp.log
for f in gpurun_out/r3d/*.json; do python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d['ms_per_step'], d['ms_per_step_all'], d['config']['final_loss'])" $f; done

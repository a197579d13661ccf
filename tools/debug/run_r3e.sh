mkdir -p gpurun_out/r3e
python -m pytest tests/test_conv_gpu.py -q -k "k4s2 or band" > gpurun_out/r3e/k4s2_ops.log 2>&1
python -m pytest tests/test_blocks_full_gpu.py tests/test_baseline_gpu.py -q -s -k "mnist_b128 and (lowp or bf16)" > gpurun_out/r3e/mnist_parity.log 2>&1
python -m pytest tests/test_eval_gpu.py -q -s > gpurun_out/r3e/eval.log 2>&1
B="python bench.py --extra_configs none --no_cpu_baseline"
VS_CONV_K4S2=0 $B --config mnist_b128 > gpurun_out/r3e/mnist_k4s2_off.json 2>/dev/null
$B --config mnist_b128 > gpurun_out/r3e/mnist_k4s2_on.json 2>gpurun_out/r3e/mnist_on.err
python bench.py --eval --config mnist_b128 --steps 5 --warmup 2 > gpurun_out/r3e/eval_mnist.json 2>gpurun_out/r3e/eval_mnist.err
VARSEP_FOLD_BN_EVAL=0 python bench.py --eval --config mnist_b128 --steps 5 --warmup 2 > gpurun_out/r3e/eval_mnist_nofold.json 2>/dev/null
python bench.py --eval --config waveeq --steps 10 --warmup 2 > gpurun_out/r3e/eval_waveeq.json 2>/dev/null
tail -3 gpurun_out/r3e/k4s2_ops.log; tail -3 gpurun_out/r3e/mnist_parity.log | cut -c1-300; tail -5 gpurun_out/r3e/eval.log | cut -c1-300
for f in gpurun_out/r3e/*.json; do python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d['value'], d['ms_per_step'], d['ms_per_step_all'])" $f; done

mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
out=gpurun_out/r3h
pmc () {   # name cfg prec extra_env...
  name=$1; cfg=$2; prec=$3
  rm -rf $out/f $out/w
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -o p -- python3 bench.py --config $cfg --precision $prec --no_graph --steps 6 --warmup 2 --repeats 1 --no_cpu_baseline --extra_configs none > $out/pmc_f_$name.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -o p -- python3 bench.py --config $cfg --precision $prec --no_graph --steps 6 --warmup 2 --repeats 1 --no_cpu_baseline --extra_configs none > $out/pmc_w_$name.log 2>&1
  python3 tools/pmc_traffic.py $(find $out/f -name "*counter_collection.csv" | head -1) $(find $out/w -name "*counter_collection.csv" | head -1) $out/r03_${name}_traffic.json $out/r03_${name}_hbm_traffic.md $name
  rm -rf $out/f $out/w
}
pmc mnist_b128_bf16 mnist_b128 bf16
export VS_CONV_IMG_PAIR=1
export VARSEP_PREPACK_CONV=0
pmc sst_bf16 sst bf16
unset VS_CONV_IMG_PAIR VARSEP_PREPACK_CONV
tail -3 $out/pmc_f_sst_bf16.log | cut -c1-200
python3 bench.py > $out/r03_bench_default.json 2> $out/bench_default.err
python3 bench.py --eval --config mnist_b128 --steps 5 --warmup 2 > $out/r03_eval_mnist_b128_bf16_bench.json 2>/dev/null
python -m pytest tests -m gpu -x -q > $out/all_gpu_tests.log 2>&1
tail -3 $out/all_gpu_tests.log
ls -la $out

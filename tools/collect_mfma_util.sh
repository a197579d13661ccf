#!/bin/bash
# MFMA / LDS / vector-memory counter evidence per kernel: four rocprofv3 --pmc passes over eager steps of a workload (counters are per
# dispatch, so the step is issued launch by launch: `bench.py --no_graph`), folded by tools/pmc_util.py.
# usage: bash tools/collect_mfma_util.sh r05 [workloads...]        (default: taxibj mnist_b128 sst)  -> gpurun_out/<tag>/<tag>_<wl>_bf16_mfma_util.{md,json}
# Each pass is wrapped in `timeout`; --pmc is never combined with a trace domain (gpurun refuses that); the program follows `--` directly.
tag=${1:-r05}
shift
wl=${@:-taxibj mnist_b128 sst}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
B="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
D="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"
for w in $wl; do
  cfg=${w%_fp16}; prec=bf16; if [ $cfg != $w ]; then prec=fp16; fi
  name=${tag}_${cfg}_${prec}
  if [ $cfg = sst ]; then export VS_CONV_IMG_PAIR=1; fi      # the PMC dispatch path rejects the 144 KiB-LDS launch (collect_profiles.sh)
  files=""
  i=0
  for set in "$A" "$B" "$C" "$D"; do
    i=$((i + 1))
    rm -rf $out/u$i
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/u$i -o p -- python3 bench.py --config $cfg --precision $prec --no_graph --steps 3 --warmup 1 --repeats 1 --no_cpu_baseline --extra_configs none > $out/util_${w}_$i.log 2>&1
    f=$(find $out/u$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then files="$files $f"; else echo "pass $i of $w produced no counter file (see $out/util_${w}_$i.log)"; tail -5 $out/util_${w}_$i.log; fi
  done
  python3 tools/pmc_util.py $out/${name}_mfma_util.md $out/${name}_mfma_util.json "$w" $files
  rm -rf $out/u1 $out/u2 $out/u3 $out/u4
  unset VS_CONV_IMG_PAIR
done
ls -la $out

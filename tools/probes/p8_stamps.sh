#!/bin/bash
# wall-clock stamps of workgroup 0 at the seams of the staggered GEMM tile (prepare | prologue | K loop | epilogue issued | stores retired), per store form:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DP8_STAMPS tools/probes/p8_bench.hip -o tools/probes/p8_bench_stamps
# output -> gpurun_out/p8_stamps.txt
cd "$(dirname "$0")/../.." || exit 2
mkdir -p gpurun_out
B=tools/probes/p8_bench_stamps
O=gpurun_out/p8_stamps.txt
: > $O
run() { echo "== $*" >> $O; timeout 120 $B "$@" >> $O 2>&1; echo "rc=$?" >> $O; }
run 256 256 64 0 0 2 rand 20
run 256 256 64 0 0 1 rand 20
run 3328 4096 1200 0 0 2 rand 20
run 3328 4096 1200 0 0 2 h16 20
run 3328 4096 1200 0 0 2 h16m 20
run 3328 4096 1200 0 0 2 loss 20
run 3328 4096 1200 0 0 2 loss_sigmoid 20
run 3328 1200 1200 0 0 1 h16 20
run 3328 1200 1200 0 0 1 h16m 20
run 3328 1200 4096 0 1 1 h16m 20
run 4096 4096 4096 0 0 2 rand 10
grep -v "(repeat)" $O

#!/bin/bash
# runs the standalone p8 GEMM harness over correctness and timing cases; output -> gpurun_out/p8_run.txt
cd "$(dirname "$0")/../.." || exit 2
mkdir -p gpurun_out
B=tools/probes/p8_bench
O=gpurun_out/p8_run.txt
: > $O
run() { echo "== $*" >> $O; timeout 300 $B "$@" >> $O 2>&1; echo "rc=$?" >> $O; }
# exact small-integer checks, R x R
for ni in 2 1; do
run 256 256 64 0 0 $ni int 5
run 256 256 128 0 0 $ni int 5
run 512 512 192 0 0 $ni int 5
run 768 512 1200 0 0 $ni int 5
run 1000 1016 1000 0 0 $ni int 5
run 3328 4096 1200 0 0 $ni int 10
run 3328 1200 1200 0 0 $ni int 10
done
# other layouts
for l in "0 1" "1 0" "1 1"; do
run 256 256 64 $l 2 int 3
run 512 768 192 $l 2 int 3
run 1000 1016 1000 $l 2 int 3
run 3328 1200 4096 $l 2 int 5
done
run 1000 1016 1000 1 0 1 int 3
# timing, uniform random operands
run 4096 4096 4096 0 0 2 rand 20
run 4096 4096 4096 0 0 1 rand 20
run 8192 8192 8192 0 0 2 rand 10
run 3328 4096 1200 0 0 2 rand 20
run 3328 4096 1200 0 0 2 cold 20
run 3328 1200 1200 0 0 1 rand 20
run 3328 1200 1200 0 0 2 rand 20
run 3328 1200 1200 0 0 1 cold 20
run 3328 1200 4096 0 1 2 rand 20
run 4096 4096 4096 0 1 2 rand 20
run 4096 4096 4096 1 0 2 rand 20
run 4096 4096 4096 1 1 2 rand 20
run 4096 1200 3328 1 1 2 rand 20
cat $O

#!/bin/bash
# counter-phase experiment of tools/ls_bench.hip: waves 4-7 run dW before dX (builds: -DLS_STAGGER=1 -DLS_BBUF=1 ;
# -DLS_BBUF=1 alone as the control), alternating on one box
out=${1:-gpurun_out/r04/ls_bench_9.log}
mkdir -p "$(dirname "$out")"; : > $out
run() { b=$1; shift; echo "\$ tools/$b $*" >> "$out"; timeout 120 tools/$b "$@" >> "$out" 2>&1; echo "rc=$?" >> "$out"; }
for b in ls_bench ls_bench_b1 ls_bench_st ls_bench ls_bench_b1 ls_bench_st; do
  run $b 524288 8 12 2 32 16 1 0
  run $b 524288 8 12 2 32 16 0 128
done
run ls_bench_st 524288 8 12 1 32 16 0 0
run ls_bench_st 524288 8 12 4 32 16 1 0
grep "^\$\|^mode\|check ok\|CHECK\|FAIL\|rc=[1-9]" "$out"

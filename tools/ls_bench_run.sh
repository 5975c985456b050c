#!/bin/bash
# Round-4 step A: the layer-stationary backward + weight-gradient micro-benchmark (tools/ls_bench.hip) on the GPU box.
# Builds the variants, runs the matrix behind profiles/r04_layer_bench_summary.md, collects the fabric counters.
#   usage: tools/ls_bench_run.sh [OUTDIR=gpurun_out/r04_ls]
dir=${1:-gpurun_out/r04_ls}
mkdir -p "$dir"
out=$dir/runs.log
: > "$out"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
build() { [ -x tools/$1 ] || $HIPCC -O3 --offload-arch=gfx950 $2 -o tools/$1 tools/ls_bench.hip; }
build ls_bench ""; build ls_bench_n3 "-DLS_NST=3"; build ls_bench_prof "-DLS_PROF"; build ls_bench_sp "-DLS_SPREAD=1"; build ls_bench_b1 "-DLS_BBUF=1"; build ls_bench_st "-DLS_STAGGER=1 -DLS_BBUF=1"
run() { b=$1; shift; echo "\$ tools/$b $*" >> "$out"; timeout 120 tools/$b "$@" >> "$out" 2>&1; echo "rc=$?" >> "$out"; }
# points layers reps mode G R check dbg
for m in 0 1 2 3; do run ls_bench 1048576 3 10 $m 85 16 1 0; done                 # 3 stages, cross-XCD pairs
for m in 2 1 4 5 3 0; do run ls_bench 524288 8 12 $m 32 16 1 0; done             # the trunk: 8 stages x 32 CUs
for d in 128 132 130 134 135; do run ls_bench 524288 8 12 2 32 16 0 $d; done       # ablations (timing only)
run ls_bench_n3 524288 8 12 4 32 8 1 0                                             # XCD-local, rings inside the L2
run ls_bench_sp 524288 8 12 2 32 16 1 0
run ls_bench_prof 524288 8 12 2 32 16 0 0                                          # cycles per phase
run ls_bench 196608 8 12 2 32 16 1 0                                               # config-2 size: the fill shows
cd /tmp && export TMPDIR=/tmp
pmc() { name=$1; ctr=$2; shift 2; rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/$dir/pmc_$name -o p -- "$@" > $GRAFT_REPO_ROOT/$dir/pmc_$name.log 2>&1; }
for c in FETCH_SIZE WRITE_SIZE; do
  pmc m1_$c $c $GRAFT_REPO_ROOT/tools/ls_bench 524288 8 3 1 32 16 0 0
  pmc m2_$c $c $GRAFT_REPO_ROOT/tools/ls_bench 524288 8 3 2 32 16 0 0
  pmc m4_R8_$c $c $GRAFT_REPO_ROOT/tools/ls_bench_n3 524288 8 3 4 32 8 0 0
done
pmc m2_hit "TCC_HIT_sum TCC_MISS_sum" $GRAFT_REPO_ROOT/tools/ls_bench 524288 8 3 2 32 16 0 0
pmc m4_R8_hit "TCC_HIT_sum TCC_MISS_sum" $GRAFT_REPO_ROOT/tools/ls_bench_n3 524288 8 3 4 32 8 0 0
cd $GRAFT_REPO_ROOT
grep "^\$\|^mode\|^prof wave\|check ok\|CHECK\|FAIL\|rc=[1-9]" "$out"

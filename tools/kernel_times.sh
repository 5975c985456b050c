#!/bin/bash
# rocprofv3 kernel durations of the replayed step under several library variants (tools/ab.sh's variant syntax):
#   tools/kernel_times.sh "label: LIB=tools/variants/x.so VAR=v" ...     -> per variant, us per step of every small kernel
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  label="${v%%:*}"; envs="${v#*:}"; envs="${envs//LIB=/HN_LIB_PATH=$GRAFT_REPO_ROOT/}"
  out=$GRAFT_REPO_ROOT/gpurun_out/kt_$label; rm -rf $out; mkdir -p $out
  env $envs rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --repeats 1 \
    --no-cpu-baseline --no-roofline --no-calibration --no-also $BENCH_ARGS > $out/line.json 2> $out/err.log
  echo "=== $label"
  python3 $GRAFT_REPO_ROOT/tools/graph_trace_summary.py $(find $out -name "*kernel_trace.csv" | head -1) | grep -A14 "per kernel name" | grep -v "hn_mlp_\|hn_wgrad_kernel"
done

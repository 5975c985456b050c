#!/bin/bash
# rocprofv3 --kernel-trace of the step replayed as ONE HIP graph (the timed path), per configuration: the independent check
# of bench.py's in-kernel timeline (verdict r05 item 2) and the ground truth for `other_ms_per_step` (every small launch's
# duration and the gaps between the nodes).   tools/graph_trace.sh rNN [config ...]
tag=${1:-r06}; shift
cfgs=${@:-2}
cd /tmp && export TMPDIR=/tmp
for cfg in $cfgs; do
  out=$GRAFT_REPO_ROOT/gpurun_out/graph_trace_${tag}_c${cfg}
  mkdir -p $out
  rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 10 --warmup 3 --repeats 1 \
    --no-cpu-baseline --no-roofline --no-calibration --no-also > $out/bench_line.json 2> $out/stderr.log
  csv=$(find $out -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/graph_trace_summary.py "$csv" $out/bench_line.json > $GRAFT_REPO_ROOT/gpurun_out/${tag}_graph_trace_config${cfg}.txt
  tail -45 $GRAFT_REPO_ROOT/gpurun_out/${tag}_graph_trace_config${cfg}.txt
done

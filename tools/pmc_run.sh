#!/bin/bash
# tools/pmc_run.sh OUTDIR "COUNTER LIST" : one rocprofv3 PMC pass over a short eager bench run
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $1 -d $out -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $out.log 2>&1
ls $out

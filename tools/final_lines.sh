#!/bin/bash
# Second pass of a round's evidence (after tools/collect_profiles.sh + tools/make_profiles.py have written the traffic
# summaries the bench line quotes): the bench lines themselves, the eval-loop bench, and the PMC bytes of the opt-in
# 8-bit-stash mode.  Writes gpurun_out/<tag>_final_*.json ; copy what is to be judged into profiles/.
tag=${1:-r03}
o=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $o/${tag}_final_config2.json
python3 bench.py --config 3 --steps 10 --warmup 3 2>/dev/null | tail -1 > $o/${tag}_final_config3.json
python3 bench.py --config 5 --steps 20 --warmup 5 2>/dev/null | tail -1 > $o/${tag}_final_config5.json
python3 bench.py --config 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $o/${tag}_final_config1.json
python3 bench.py --precision fp32 --steps 10 --warmup 3 2>/dev/null | tail -1 > $o/${tag}_final_config2_fp32.json
python3 bench.py --precision bf16s8 --steps 20 --warmup 5 2>/dev/null | tail -1 > $o/${tag}_final_config2_s8.json
for c in 8192 32768 65536; do python3 tools/eval_bench.py $c 5 2>/dev/null | tail -1; done > $o/${tag}_final_eval.jsonl
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
COMMON="--precision bf16s8 --repeats 1 --no-cpu-baseline --no-roofline --no-graph"
mkdir -p $o/prof_${tag}_c2s8
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/prof_${tag}_c2s8/pmc_fetch -o p -- python3 $B --steps 3 --warmup 2 $COMMON > $o/prof_${tag}_c2s8/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $o/prof_${tag}_c2s8/pmc_write -o p -- python3 $B --steps 3 --warmup 2 $COMMON > $o/prof_${tag}_c2s8/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag}_c2s8/stats -o s -- python3 $B --steps 20 --warmup 3 $COMMON > $o/prof_${tag}_c2s8/stats.log 2>&1
ls $o | grep ${tag}_final

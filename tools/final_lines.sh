#!/bin/bash
# Second pass of a round's evidence (after tools/collect_profiles.sh + tools/make_profiles.py have written the traffic
# summaries the bench line quotes): the bench lines themselves (the config-2 line WITH its `also` block, as the driver
# runs it), the other configurations as full lines, the fp32 parity mode and the eval-loop bench.
# Writes gpurun_out/<tag>_final_*.json ; copy what is to be judged into profiles/ (tools/finish_profiles.py).
# (Round 3 also collected the opt-in 8-bit-stash mode here; that mode is frozen since round 4: FINAL_S8=1 brings it back.)
tag=${1:-r05}
o=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python3 bench.py 2>/dev/null | tail -1 > $o/${tag}_final_config2.json
python3 bench.py --config 3 --steps 10 --warmup 3 2>/dev/null | tail -1 > $o/${tag}_final_config3.json
python3 bench.py --config 5 --steps 20 --warmup 5 2>/dev/null | tail -1 > $o/${tag}_final_config5.json
python3 bench.py --config 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $o/${tag}_final_config1.json
python3 bench.py --precision fp32 --steps 10 --warmup 3 --no-also 2>/dev/null | tail -1 > $o/${tag}_final_config2_fp32.json
python3 bench.py --force-dp --no-also --no-cpu-baseline 2>/dev/null | tail -1 > $o/${tag}_final_config2_dp1.json
for c in 8192 32768 65536; do python3 tools/eval_bench.py $c 5 2>/dev/null | tail -1; done > $o/${tag}_final_eval.jsonl
if [ "$FINAL_S8" = "1" ]; then
  python3 bench.py --precision bf16s8 --steps 20 --warmup 5 2>/dev/null | tail -1 > $o/${tag}_final_config2_s8.json
fi
ls $o | grep ${tag}_final

#!/bin/bash
# ONE same-box A/B tool (round 6: replaces ab_env*.sh, ab_multi.sh, ab_prebuilt.sh, ab_build2.sh, ab_src.sh, ab_reuse.sh,
# ab_check.sh, ab_test.sh, ab_wgrad_exp.sh).  Variants run round-robin in separate processes on the box the call got:
#
#   tools/ab.sh [-n ROUNDS] [-b "bench.py args"] [-k "pytest -k expr"] [-o LOG] "label: VAR=v LIB=path.so ..." ...
#
# A variant is a label and a list of environment assignments.  LIB=path selects a library PREBUILT in the build container
# (tools/build_variant.sh; it travels with the snapshot) through HN_LIB_PATH — the product library is never overwritten
# (advisor, round 5).  Run-time knobs (HN_FUSE_REDUCE=0, HN_REUSE_COARSE=0, HN_WGRAD_STAGE_KB=48, ...) go in as they are.
# -k: first run that parity subset of the GPU suite under every variant (a variant's time is only read once it passes).
# Output: one line per run — ms/step, M ray-samples/s, the machine kernels' timeline and `other` — and the medians.
ROUNDS=3; BARGS=""; KEXPR=""; LOG=""
while getopts "n:b:k:o:" o; do case $o in n) ROUNDS=$OPTARG;; b) BARGS=$OPTARG;; k) KEXPR=$OPTARG;; o) LOG=$OPTARG;; esac; done
shift $((OPTIND - 1))
cd "$(dirname "$0")/.."
[ -n "$LOG" ] && exec > >(tee "$LOG") 2>&1
envs_of() { local v="${1#*:}"; echo "${v//LIB=/HN_LIB_PATH=}"; }
if [ -n "$KEXPR" ]; then
  for v in "$@"; do
    echo "=== check ${v%%:*}"
    env $(envs_of "$v") timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" 2>&1 | tail -2
  done
fi
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    env $(envs_of "$v") timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --no-calibration $BARGS 2>/dev/null | tail -1 | \
      AB_LABEL="${v%%:*}" python tools/ab_line.py
  done
done | tee /tmp/ab_lines.$$
python tools/ab_line.py --summary /tmp/ab_lines.$$
rm -f /tmp/ab_lines.$$

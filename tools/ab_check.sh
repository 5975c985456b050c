#!/bin/bash
# parity subset under a (prebuilt library, environment) pair: tools/ab_check.sh "lib.so VAR=val" ...
LIB=hypernerf-torch_amd/csrc/libhn_hip.so
cp $LIB /tmp/lib_keep.so
for cfg in "$@"; do
  lib=${cfg%% *}; envs=""; [ "$lib" != "$cfg" ] && envs=${cfg#* }
  echo "=== check $cfg"
  cp "$lib" $LIB
  env $envs timeout 600 python -m pytest tests -m gpu -x -q -k "${CHECK_K:-partial_slabs or g03 or bf16_operand or golden_model or fuzz}" 2>&1 | tail -3
done
cp /tmp/lib_keep.so $LIB

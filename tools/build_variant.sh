#!/bin/bash
# Build an A/B variant of the library BESIDE the product one (in the build container; the .so travels with the snapshot):
#   tools/build_variant.sh tools/variants/stages3.so HN_WGRAD_STAGES=3 HN_WGRAD_MAXSLOT=6
# optionally from another source file:  SRC_MLP=path/to/hn_mlp_variant.hip tools/build_variant.sh out.so
# tools/ab.sh "label: LIB=tools/variants/stages3.so" then runs it through HN_LIB_PATH.
out=$1; shift
cd "$(dirname "$0")/.."
mkdir -p "$(dirname "$out")"
csrc=hypernerf-torch_amd/csrc
defs=""; for kv in "$@"; do defs="$defs -D$kv"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fPIC -shared $defs \
  -I$csrc -o "$out" ${SRC_MLP:-$csrc/hn_mlp.hip} ${SRC_RENDER:-$csrc/hn_render.hip} $csrc/hn_calib.hip && echo "built $out ($*)"

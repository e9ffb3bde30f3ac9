#!/bin/bash
# builds an A/B variant of the HIP library: tools/build_variant.sh <name> [extra hipcc flags...]  ->  lib/libfrirl_hip_<name>.so
# (load it with FRIRL_HIP_LIB_OVERRIDE=<path>; experiments only, never shipped)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/fri-reinforcementlearning-c_amd
name=$1; shift
obj=$PKG/build/variant_$name
mkdir -p "$obj" "$PKG/lib"
pids=()
for src in "$PKG"/csrc/*.hip; do
  o=$obj/$(basename "${src%.hip}").o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I "$ROOT/include" "$@" -c "$src" -o "$o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/lib/libfrirl_hip_$name.so" "$obj"/*.o -ldl -lpthread
echo "built $PKG/lib/libfrirl_hip_$name.so"

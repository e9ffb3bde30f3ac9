#!/bin/bash
# variant of libfrirl_hip.so with extra -D flags for the learner TUs only:  tools/exp/build_learn_variant.sh NAME -DLEARN_TIMING ...
set -e
cd "$(dirname "$0")/../../fri-reinforcementlearning-c_amd"
name=$1; shift
mkdir -p build/variant_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I ../include"
for f in learn learn_i0 learn_i1 learn_i2; do /opt/rocm/bin/hipcc $FLAGS "$@" -c csrc/$f.hip -o build/variant_$name/$f.o & done; wait
objs=""; for o in build/*.o; do b=$(basename $o); if [ -f build/variant_$name/$b ]; then objs="$objs build/variant_$name/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libfrirl_hip_$name.so $objs -ldl -lpthread
echo built lib/libfrirl_hip_$name.so

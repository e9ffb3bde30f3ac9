#!/bin/bash
# CPU AddressSanitizer + UBSan runs (GPU sanitizers are not available on the pool):
#  - the oracle: three demo runs, reductions, synthetic generators, batched distance
#  - the host-only part of the drop-in library (table generation: FIVE_GSc_func, FIVEGVagEnv, frirl_init_ve)
set -e
cd "$(dirname "$0")/../.."
mkdir -p /tmp/frirl_asan
gcc -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fopenmp -Ioracle tools/asan/oracle_asan.c oracle/frirl_oracle.c -o /tmp/frirl_asan/oracle -lm
ASAN_OPTIONS=detect_leaks=1 /tmp/frirl_asan/oracle
H=fri-reinforcementlearning-c_amd/host
gcc -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -std=gnu99 -Iinclude -I$H tools/asan/host_asan.c $H/five_host.c $H/frirl_host.c $H/frirl_io.c $H/demo_envs.c \
    -o /tmp/frirl_asan/host -Lfri-reinforcementlearning-c_amd/lib -lfrirl_hip -Wl,-rpath,$PWD/fri-reinforcementlearning-c_amd/lib -lm
/tmp/frirl_asan/host
echo "asan/ubsan: clean"

#!/bin/bash
# scratch/build_variant_pnp.sh NAME [-DFLAG ...]: scratch/variants/libesfm_NAME.so with pnp_kernels.hip recompiled under the flags
set -e
name=$1; shift
cd /root/repo/easysfm_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function "$@" -x hip -c pnp_kernels.hip -o /tmp/pk_$name.o
objs=$(ls build/*.o | grep -v pnp_kernels)
mkdir -p ../../scratch/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/variants/libesfm_$name.so $objs /tmp/pk_$name.o -ldl

#!/bin/bash
# scratch/build_variant_ba.sh NAME [-DFLAG ...]: scratch/variants/libesfm_NAME.so with ba_kernels.hip recompiled under the flags
set -e
name=$1; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}/easysfm_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function "$@" -x hip -c ba_kernels.hip -o /tmp/bk_$name.o
objs=$(ls build/*.o | grep -v ba_kernels)
mkdir -p ../../scratch/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/variants/libesfm_$name.so $objs /tmp/bk_$name.o -ldl

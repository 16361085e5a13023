#!/bin/bash
# scratch/build_variant.sh NAME [-DFLAG ...]: builds scratch/variants/libesfm_NAME.so with match_kernels.hip recompiled under the flags
set -e
name=$1; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}/easysfm_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form "$@" -x hip -c match_kernels.hip -o /tmp/mk_$name.o
mkdir -p ../../scratch/variants
objs=$(ls build/*.o | grep -v match_kernels)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/variants/libesfm_$name.so $objs /tmp/mk_$name.o

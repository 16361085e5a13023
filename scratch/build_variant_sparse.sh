#!/bin/bash
# scratch/build_variant_sparse.sh NAME [-DFLAG ...]: scratch/variants/libesfm_NAME.so with ba_chol_sparse.hip recompiled under the flags
set -e
name=$1; shift
cd /root/repo/easysfm_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function "$@" -x hip -c ba_chol_sparse.hip -o /tmp/sp_$name.o
objs=$(ls build/*.o | grep -v ba_chol_sparse)
mkdir -p ../../scratch/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/variants/libesfm_$name.so $objs /tmp/sp_$name.o -ldl

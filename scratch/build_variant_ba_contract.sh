#!/bin/bash
# like scratch/build_variant_ba.sh but with the contraction flag replaceable: bv.sh NAME CONTRACT [-D...]
set -e
name=$1; con=$2; shift; shift
cd /root/repo/easysfm_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=$con -munsafe-fp-atomics -Wall -Wno-unused-function "$@" -x hip -c ba_kernels.hip -o /tmp/bk_$name.o
objs=$(ls build/*.o | grep -v ba_kernels)
mkdir -p ../../scratch/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/variants/libesfm_$name.so $objs /tmp/bk_$name.o -ldl

#!/bin/bash
# timing-only builds of hamming_fp4_kernel: the generated main loop with the fold / the MFMAs compiled out (wrong results)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT/easysfm_amd/csrc
cp hmx1_segment_gfx950.inc /tmp/hmx1_segment_gfx950.inc.orig
# (the variants produce WRONG results: whatever happens below, the tracked inner loop comes back)
trap 'cp /tmp/hmx1_segment_gfx950.inc.orig '"$ROOT"'/easysfm_amd/csrc/hmx1_segment_gfx950.inc' EXIT
G="ESFM_GEN_MFMA=fp4 ESFM_GEN_KEEP=2 ESFM_GEN_STEP_BITS=13 ESFM_GEN_PREFIX=ESFM_HMX1"
for v in nofold:ESFM_GEN_NOFOLD=1 nomfma:ESFM_GEN_NOMFMA=1 neither:"ESFM_GEN_NOFOLD=1 ESFM_GEN_NOMFMA=1"; do
  name=${v%%:*}; knob=${v#*:}
  env $G $knob python3 gen_l2x1_segment_asm.py hmx1_segment_gfx950.inc
  bash ../../scratch/build_variant.sh hm_$name
done
cp /tmp/hmx1_segment_gfx950.inc.orig hmx1_segment_gfx950.inc
git -C $ROOT diff --stat -- easysfm_amd/csrc/hmx1_segment_gfx950.inc

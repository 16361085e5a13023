#!/bin/bash
# config 1 / config 3 through bin/sfm_native, per-frame trace (ESFM_FRAME_TRACE): undistort / detect milliseconds of every frame.
# History: HSA_ENABLE_SDMA=0 did not remove the 20 - 38 ms stalls of some frames, GPU_PINNED_MIN_XFER_SIZE=128 did (the runtime pins
# pageable buffers of 1 MiB and more): the library now copies in 512-KiB pieces (common.hpp copy_h2d / copy_d2h).
cd $GRAFT_REPO_ROOT
O=gpurun_out/frame_trace; rm -rf $O; mkdir -p $O/data/images_25 $O/data/k_25
python3 - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, '.')
import bench
z = np.load('tests/golden/fountain11_gray.npz')['images']
names = []
for i, im in enumerate(z):
    names.append(f'{i:04d}.png')
    bench._write_png_rgb(f'gpurun_out/frame_trace/data/images_25/{names[-1]}', np.ascontiguousarray(np.stack([im] * 3, axis=2)))
open('gpurun_out/frame_trace/data/image_list.txt', 'w').write('\n'.join(names) + '\n')
open('gpurun_out/frame_trace/data/k_25/K.txt', 'w').write('689.87 0 380.17\r\n0 691.04 251.70\r\n0 0 1')
PY
D=$O/data
for rep in 1 2 3; do
for S in default; do
for F in "S 300" "O 8000"; do
  echo "== feature $F (undistort ms / detect ms per frame)"
  ESFM_FRAME_TRACE=1 ./bin/sfm_native $D/images_25 $D/image_list.txt $D/k_25/K.txt none $O/out.ply $F 1.0 1 0 4 0 0 2> $O/err.txt > $O/log.txt
  grep -E "^\[frame" $O/err.txt | awk '{printf "%.1f/%.1f ", $8, $11} END {print ""}'
  grep "stage seconds" $O/log.txt | cut -c1-90
done; done; done
rm -rf $O/data

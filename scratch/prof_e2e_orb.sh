#!/bin/bash
# kernel trace of bin/sfm_native on the fountain images (config 1): which kernels the end-to-end run spends its GPU time in
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/prof_e2e_orb; rm -rf $O; mkdir -p $O/data/images_25 $O/data/k_25
python3 - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, '.')
import bench
z = np.load('tests/golden/fountain11_gray.npz')['images']
names = []
for i, im in enumerate(z):
    names.append(f'{i:04d}.png')
    bench._write_png_rgb(f'gpurun_out/prof_e2e_orb/data/images_25/{names[-1]}', np.ascontiguousarray(np.stack([im] * 3, axis=2)))
open('gpurun_out/prof_e2e_orb/data/image_list.txt', 'w').write('\n'.join(names) + '\n')
open('gpurun_out/prof_e2e_orb/data/k_25/K.txt', 'w').write('689.87 0 380.17\r\n0 691.04 251.70\r\n0 0 1')
PY
D=$O/data
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- ./bin/sfm_native $D/images_25 $D/image_list.txt $D/k_25/K.txt none $O/out.ply O 8000 1.0 1 0 4 0 0 > $O/log.txt 2>&1
python3 tools/rocprof_csv_summary.py $O/t | head -24 | cut -c1-150
grep "stage seconds" $O/log.txt
find $O -name "*.csv" -size +2M -delete; rm -rf $O/data

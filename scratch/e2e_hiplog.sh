#!/bin/bash
# config 1 through bin/sfm_native with the HIP runtime's API log: where do the 20-ms stalls of some frames' detect calls come from
cd $GRAFT_REPO_ROOT
O=gpurun_out/hiplog; rm -rf $O; mkdir -p $O/data/images_25 $O/data/k_25
python3 - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, '.')
import bench
z = np.load('tests/golden/fountain11_gray.npz')['images']
names = []
for i, im in enumerate(z):
    names.append(f'{i:04d}.png')
    bench._write_png_rgb(f'gpurun_out/hiplog/data/images_25/{names[-1]}', np.ascontiguousarray(np.stack([im] * 3, axis=2)))
open('gpurun_out/hiplog/data/image_list.txt', 'w').write('\n'.join(names) + '\n')
open('gpurun_out/hiplog/data/k_25/K.txt', 'w').write('689.87 0 380.17\r\n0 691.04 251.70\r\n0 0 1')
PY
D=$O/data
ESFM_FRAME_TRACE=1 AMD_LOG_LEVEL=3 ./bin/sfm_native $D/images_25 $D/image_list.txt $D/k_25/K.txt none $O/out.ply S 300 1.0 1 0 4 0 0 > $O/log.txt 2> $O/err.txt
grep "^\[frame" $O/err.txt
python3 - <<'PY'
import re
prev = None
for l in open('gpurun_out/hiplog/err.txt', errors='replace'):
    m = re.search(r'\[pid:\d+ tid:\s*(0x[0-9a-f]+)\]', l)
    t = re.match(r':\d+:[^:]*:\s*(\d+)\s*:\s*(\d+) us', l)
    if not t: continue
    ts = int(t.group(2))
    if prev is not None and ts - prev[0] > 5000:
        print('GAP', (ts - prev[0]) / 1000, 'ms after:', prev[1][:200].rstrip()); print('     then:', l[:200].rstrip())
    prev = (ts, l)
PY
head -c 3000 $O/err.txt | head -20
rm -rf $O/data; find $O -size +20M -delete

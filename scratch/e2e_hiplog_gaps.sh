#!/bin/bash
# config 1 through bin/sfm_native with the HIP API log: every API call that took more than 0.3 ms, and gaps between log lines above 1 ms
cd $GRAFT_REPO_ROOT
O=gpurun_out/hiplog2; rm -rf $O; mkdir -p $O/data/images_25 $O/data/k_25
python3 - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, '.')
import bench
z = np.load('tests/golden/fountain11_gray.npz')['images']
names = []
for i, im in enumerate(z):
    names.append(f'{i:04d}.png')
    bench._write_png_rgb(f'gpurun_out/hiplog2/data/images_25/{names[-1]}', np.ascontiguousarray(np.stack([im] * 3, axis=2)))
open('gpurun_out/hiplog2/data/image_list.txt', 'w').write('\n'.join(names) + '\n')
open('gpurun_out/hiplog2/data/k_25/K.txt', 'w').write('689.87 0 380.17\r\n0 691.04 251.70\r\n0 0 1')
PY
D=$O/data
ESFM_FRAME_TRACE=1 AMD_LOG_LEVEL=3 ./bin/sfm_native $D/images_25 $D/image_list.txt $D/k_25/K.txt none $O/out.ply S 300 1.0 1 0 4 0 0 > $O/log.txt 2> $O/err.txt
grep "stage seconds" $O/log.txt | cut -c1-250
python3 - <<'PY'
import re
prev = None; last_kernel = ''
for l in open('gpurun_out/hiplog2/err.txt', errors='replace'):
    t = re.match(r':\d+:[^:]*:\s*(\d+)\s*:\s*(\d+) us', l)
    if 'ShaderName' in l: last_kernel = l.split('ShaderName :')[1][:50].strip()
    if not t: continue
    ts = int(t.group(2))
    d = re.search(r'(hip\w+): Returned \w+ :.*duration: (\d+) us', l)
    if d and int(d.group(2)) > 300: print(f"{ts} us: {d.group(1)} took {int(d.group(2))} us (last kernel enqueued: {last_kernel})")
    if prev is not None and ts - prev > 1000 and not d: print(f"{ts} us: gap {ts - prev} us before: {re.sub(chr(27) + r'.[0-9;]*m', '', l[40:140]).strip()}")
    prev = ts
PY
rm -rf $O/data; find $O -size +20M -delete

#!/bin/bash
# HIP API timeline of ONE steady-state esfm_orb_detect_and_compute call (768 x 512, 8000 features): where the ~1 ms goes on the host
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/orblog
AMD_LOG_LEVEL=3 python3 - > gpurun_out/orblog/out.txt 2> gpurun_out/orblog/err.txt <<'PY'
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
z = np.load("tests/golden/fountain11_gray.npz")["images"]
ctx = E.Context(0, None)
bgr = np.ascontiguousarray(np.stack([z[3]] * 3, axis=2))
for _ in range(3): E.orb_detect_and_compute(bgr, 8000, None, ctx)
sys.stderr.write("MARK-BEGIN\n"); sys.stderr.flush()
t0 = time.perf_counter(); E.orb_detect_and_compute(bgr, 8000, None, ctx); el = time.perf_counter() - t0
sys.stderr.write("MARK-END\n"); sys.stderr.flush()
print("call ms", el * 1e3)
PY
cat gpurun_out/orblog/out.txt
python3 - <<'PY'
import re
lines = open('gpurun_out/orblog/err.txt', errors='replace').read().split('MARK-BEGIN')[-1].split('MARK-END')[0].splitlines()
prev = None; t_first = None
for l in lines:
    t = re.match(r':\d+:[^:]*:\s*(\d+)\s*:\s*(\d+) us', l)
    if not t: continue
    ts = int(t.group(2))
    if t_first is None: t_first = ts
    m = re.search(r'(hip\w+) \(', l)
    d = re.search(r'(hip\w+): Returned \w+ :.*?(?:duration: (\d+) us)?$', l)
    if m: print(f"{ts - t_first:7d} us  call {m.group(1)}", re.sub(r'\x1b\[[0-9;]*m', '', l.split(m.group(1))[1])[:90])
    if 'ShaderName' in l: print(f"{ts - t_first:7d} us     kernel", l.split('ShaderName :')[1][:70])
PY
find gpurun_out/orblog -size +5M -delete

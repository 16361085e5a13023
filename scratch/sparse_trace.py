# timeline of chol_sparse_kernel's workgroups (variant built with -DESFM_SPARSE_TRACE; run with ESFM_LIB=scratch/variants/libesfm_sptrace.so)
import sys, ctypes, os; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
nc, npt, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 300000, 10)))
sc = synth.ba_scene(nc, npt, k, radius=40.0 if nc >= 400 else 15.0, extent=8.0 if nc >= 400 else 3.0, seed=5000)
ctx = E.Context(0)
prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
opt = E.default_options(); opt.max_num_iterations = 4; opt.function_tolerance = 0; opt.parameter_tolerance = 0
prob.solve(opt); ctx.synchronize()
lib = _lib.lib()
buf = (ctypes.c_ulonglong * (4096 * 8))()
rc = lib.esfm_debug_sparse_trace(buf)
T = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
plan = E.reduced_plan(nc, npt, sc.cam_idx, sc.pt_idx)
# rebuild the workgroup list as the library does: column by column; kinds from the tile list (needs the same order: see ba_sparse_plan.cpp)
tiles = plan['tiles']; nb = plan['nb']
L = np.zeros((nb + 1, nb), bool); L[tiles[:, 0], tiles[:, 1]] = True
klast = [max([k for k in range(i) if L[i, k]], default=-1) for i in range(nb)]
wgs = []
for j in range(nb):
    if klast[j] < 0: wgs.append((j, j, 1))
    for i in range(j + 1, nb + 1):
        if L[i, j]: wgs.append((i, j, 3 if i == nb else (2 if klast[i] == j else 0)))
assert len(wgs) == plan['workgroups'], (len(wgs), plan['workgroups'])
t0 = T[:len(wgs), 0].min()
us = lambda x: (x - t0) * 10e-3
print(f"rc {rc}; nb {nb}, chain {plan['chain']}, {len(wgs)} workgroups; kernel span {us(T[:len(wgs)].max()):.1f} us")
print(" wg   (I, J) kind |  start  upd-done (unused)  inv-seen  diag-ready  x-pub  factor-done")
ready = {}
for b, (i, j, kind) in enumerate(wgs):
    r = T[b]
    if kind in (1, 2):
        ready[i if kind == 2 else j] = r[7]
for b, (i, j, kind) in enumerate(wgs):
    r = T[b]
    if kind in (1, 2) or '-v' in sys.argv:
        f = lambda q: f"{us(r[q]):8.1f}" if r[q] else "       -"
        print(f"{b:4d} ({i:3d},{j:3d})  {kind}   | {f(0)} {f(1)} {f(2)} {f(3)} {f(4)} {f(5)} {f(7)}")
cols = sorted(ready, key=lambda c: ready[c])
print("columns in the order their inverses were published:", [(c, round(us(ready[c]), 1)) for c in cols])

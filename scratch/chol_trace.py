# stage times of chol3_kernel's chain workgroups (variant built with -DESFM_CHOL_TRACE; run with ESFM_LIB=scratch/variants/libesfm_choltrace.so)
import sys, ctypes; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
sc = synth.ba_scene(512, 300000, 10, radius=40.0, extent=8.0, seed=5000)
ctx = E.Context(0)
prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
opt = E.default_options(); opt.max_num_iterations = 4; opt.function_tolerance = 0; opt.parameter_tolerance = 0
prob.solve(opt); ctx.synchronize()
lib = _lib.lib()
buf = (ctypes.c_ulonglong * (64 * 12))()
rc = lib.esfm_debug_chol_trace(buf)
full = np.frombuffer(buf, dtype=np.uint64).reshape(64, 12).astype(np.int64)
cyc = np.median(full[5:46, 8]); full[:, 8] = full[:, 7]
t = full[:47, :10]
print('the 4 pivot chains in shader cycles (s_memtime): %.0f  ->  %.2f GHz while they run' % (cyc, cyc / (np.median(full[5:46, 10]) * 10.0) ))
print('inside the tile factorisation: 4 pivot chains (potrf16) %.2f us, everything between them %.2f us' % (np.median(full[5:46, 10]) * 10e-3, np.median(full[5:46, 11]) * 10e-3))
names = ['start', 'loop done', 'dpart got', 'ready seen', 'Linv in LDS', 'X + diag upd', 'xcount pub', 'potrf64', 'inv64', 'published']
d = np.diff(t, axis=1) * 10e-3          # us
print('rc', rc, ' per-stage us (median over columns 5..45):')
for q in range(1, 10):
    print(f'  {names[q - 1]:>13s} -> {names[q]:<13s} {np.median(d[5:46, q - 1]):7.2f}   (min {d[5:46, q - 1].min():6.2f} max {d[5:46, q - 1].max():6.2f})')
per = np.diff(t[:, 9]) * 10e-3
print('column period (published -> published): median %.2f us, total %.1f us' % (np.median(per[5:45]), (t[46, 9] - t[0, 9]) * 10e-3))
hand = (t[1:, 3] - t[:-1, 9]) * 10e-3
print('hand-over published(j) -> ready seen(j+1): median %.2f us' % np.median(hand[5:45]))

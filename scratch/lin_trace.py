"""Stage times of ba_linearize_kernel on BA-512 (build: scratch/build_variant_ba.sh lintr -DESFM_LIN_TRACE; ESFM_LIB=scratch/variants/libesfm_lintr.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easysfm_amd as E
from easysfm_amd import synth, _lib
sc = synth.ba_scene(25, 30000, 8, radius=10.0, extent=2.0, seed=4000) if len(sys.argv) > 1 and sys.argv[1] == "ba25" else synth.ba_scene(512, 300000, 10, radius=40.0, extent=8.0, seed=5000)
ctx = E.Context(0)
prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
opt = E.default_options(); opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
opt.max_num_iterations = 4
prob.solve(opt)
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
# continue from the current parameters (no set_params: the first sweeps of a solve are the two-pass form)
lib.esfm_debug_lin_trace(out, 1)
opt.max_num_iterations = 10
ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_BA_LINEARIZE)
s = prob.solve(opt); ctx.synchronize()
ts = ctx.kernel_time(_lib.K_BA_LINEARIZE)
lib.esfm_debug_lin_trace(out, 0)
# the waves of the LAST sweep launch: when each started, looped, ended; per XCC
W = (ctypes.c_ulonglong * (512 * 8 * 5))()
lib.esfm_debug_lin_waves(W)
w = np.array(list(W), dtype=np.int64).reshape(-1, 5)
w = w[w[:, 0] > 0]
t_first = w[:, 0].min()
st, ls, le, en, xcc = ((w[:, k] - t_first) * 0.01 for k in range(4)) if False else (None,) * 5
st = (w[:, 0] - t_first) * 0.01; ls = (w[:, 1] - t_first) * 0.01; le = (w[:, 2] - t_first) * 0.01; en = (w[:, 3] - t_first) * 0.01; xcc = w[:, 4]
print(f"last launch: {len(w)} waves; start {st.min():.1f} .. {st.max():.1f} us, loop start {ls.min():.1f} .. {ls.max():.1f}, loop end {le.min():.1f} .. {le.max():.1f}, end {en.min():.1f} .. {en.max():.1f}")
loop = le - ls
print(f"loop time per wave: min {loop.min():.1f} p10 {np.percentile(loop, 10):.1f} median {np.median(loop):.1f} p90 {np.percentile(loop, 90):.1f} max {loop.max():.1f} us")
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"  xcc {x}: {m.sum()} waves, start {st[m].min():.1f}..{st[m].max():.1f}, loop median {np.median(loop[m]):.1f} max {loop[m].max():.1f}, end max {en[m].max():.1f}")

# between workgroups (CUs) or within them?
idx = np.nonzero(np.array(list(W), dtype=np.int64).reshape(-1, 5)[:, 0] > 0)[0]
wg = idx // 8
per_wg = np.array([loop[wg == g].mean() for g in sorted(set(wg.tolist()))])
within = np.array([loop[wg == g].max() - loop[wg == g].min() for g in sorted(set(wg.tolist()))])
print(f"workgroup means: min {per_wg.min():.1f} median {np.median(per_wg):.1f} max {per_wg.max():.1f} us; spread inside a workgroup: median {np.median(within):.1f} max {within.max():.1f} us")
order = np.argsort(per_wg)
print("slowest workgroups (index: mean loop us, xcc):", [(int(sorted(set(wg.tolist()))[g]), round(float(per_wg[g]), 1), int(xcc[wg == sorted(set(wg.tolist()))[g]][0])) for g in order[-8:]])
print("fastest workgroups:", [(int(sorted(set(wg.tolist()))[g]), round(float(per_wg[g]), 1), int(xcc[wg == sorted(set(wg.tolist()))[g]][0])) for g in order[:8]])

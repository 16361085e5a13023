# BA-512 with the points renumbered by lowest camera BEFORE the problem is created: what would storing the linearisation in chunk order buy?
import sys, time; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
sc = synth.ba_scene(512, 300000, 10, radius=40.0, extent=8.0, seed=5000)
ctx = E.Context(0)
def run(cam_idx, pt_idx, uv, pts0, label):
    prob = E.BAProblem(cam_idx, pt_idx, uv, sc.K4, sc.cams0, pts0, ctx)
    opt = E.default_options(); opt.function_tolerance = 0; opt.parameter_tolerance = 0; opt.gradient_tolerance = 0
    opt.max_num_iterations = 3; prob.solve(opt)
    best = 0
    for rep in range(3):
        prob.set_params(sc.cams0, pts0); opt.max_num_iterations = 12
        ctx.synchronize(); t = time.perf_counter(); s = prob.solve(opt); ctx.synchronize(); best = max(best, s.num_iterations / (time.perf_counter() - t))
    prob.set_params(sc.cams0, pts0); ctx.set_kernel_timing(True)
    for k in (_lib.K_BA_LINEARIZE, _lib.K_BA_SCHUR, _lib.K_BA_SOLVE): ctx.kernel_time(k)
    s = prob.solve(opt); ctx.synchronize()
    out = {}
    for k, n in ((_lib.K_BA_LINEARIZE, 'linearize'), (_lib.K_BA_SCHUR, 'schur'), (_lib.K_BA_SOLVE, 'solve')):
        ms, c = ctx.kernel_time(k); out[n] = round(ms / max(c, 1) * 1e3, 1)
    ctx.set_kernel_timing(False)
    print(f'{label}: {best:7.1f} it/s kernels us {out} final cost {s.final_cost:.6f}')
    prob.close()
run(sc.cam_idx, sc.pt_idx, sc.uv, sc.pts0, 'as generated ')
# rotated lowest camera like the library (plain numbering is enough here)
lo = np.full(sc.n_pt, 1 << 30, np.int64); np.minimum.at(lo, sc.pt_idx, sc.cam_idx)
order = np.argsort(lo, kind='stable'); rank = np.empty_like(order); rank[order] = np.arange(sc.n_pt)
run(sc.cam_idx, rank[sc.pt_idx].astype(np.int32), sc.uv, sc.pts0[order], 'points sorted')

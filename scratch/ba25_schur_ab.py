import os, sys, time; sys.path.insert(0, '.')
import numpy as np, torch, easysfm_amd as E
from easysfm_amd import synth, _lib
sc = synth.ba_scene(25, 30000, 8, radius=10.0, extent=2.0, seed=4000)
ctx = E.Context.on_torch_stream(0)
with torch.cuda.stream(ctx.torch_stream):
    prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
    opt = E.default_options(); opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
    opt.max_num_iterations = 3; prob.solve(opt)
    best = 0.0
    for rep in range(5):
        prob.set_params(sc.cams0, sc.pts0); opt.max_num_iterations = 50
        ctx.synchronize(); t0 = time.perf_counter(); s = prob.solve(opt); ctx.synchronize()
        best = max(best, s.num_iterations / (time.perf_counter() - t0))
    prob.set_params(sc.cams0, sc.pts0); ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_BA_SCHUR)
    s = prob.solve(opt); ctx.synchronize(); k = ctx.kernel_time(_lib.K_BA_SCHUR)
print(f"ESFM_BA_SCHUR={os.environ.get('ESFM_BA_SCHUR', '-'):5s} BA-25 {best:8.1f} LM it/s  schur launches avg {k[0] / max(k[1], 1) * 1e3:.1f} us  final cost {s.final_cost:.9f} accepted {s.num_successful_steps}")

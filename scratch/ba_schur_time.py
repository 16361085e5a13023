import sys, time; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
sc = synth.ba_scene(512, 300000, 10, radius=40.0, extent=8.0, seed=5000)
ctx = E.Context(0)
prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
opt = E.default_options(); opt.function_tolerance = 0; opt.parameter_tolerance = 0; opt.gradient_tolerance = 0
opt.max_num_iterations = 2; prob.solve(opt); prob.set_params(sc.cams0, sc.pts0)
ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_BA_SCHUR)
opt.max_num_iterations = 6
try:
    prob.solve(opt)
except Exception as e:
    print('solve raised', repr(e)[:80])
ms, c = ctx.kernel_time(_lib.K_BA_SCHUR); print('schur avg us', ms / max(c, 1) * 1e3, 'over', c)

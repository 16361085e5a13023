# linearize kernel time on BA-25 and BA-512 (first solve's launches; ESFM_EXP_LIN breaks the solve, timing only)
import sys, time; sys.path.insert(0,'.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
for name, sc in (("ba25", synth.ba_scene(25, 30000, 8, radius=10.0, extent=2.0, seed=4000)), ("ba512", synth.ba_scene(512, 300000, 10, radius=40.0, extent=8.0, seed=5000))):
    ctx = E.Context(0)
    prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
    opt = E.default_options(); opt.max_num_iterations = 4; opt.function_tolerance=0; opt.parameter_tolerance=0; opt.gradient_tolerance=0
    for rep in range(2):
        prob.set_params(sc.cams0, sc.pts0)
        ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_BA_LINEARIZE)
        try:
            prob.solve(opt)
        except Exception as e:
            print('solve failed', repr(e)[:80])
        ms, c = ctx.kernel_time(_lib.K_BA_LINEARIZE)
        ctx.set_kernel_timing(False)
    print(name, 'linearize ms', ms / max(c, 1), 'launches', c)

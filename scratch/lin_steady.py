"""Steady-state time of the Jacobian sweep (ba_linearize_kernel) and LM it/s on BA-25 and BA-512 for the library in ESFM_LIB: 20 LM
iterations, kernel timers on in a second solve.  One line per problem."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import easysfm_amd as E
from easysfm_amd import synth, _lib
for name, sc in (("ba25", synth.ba_scene(25, 30000, 8, radius=10.0, extent=2.0, seed=4000)), ("ba512", synth.ba_scene(512, 300000, 10, radius=40.0, extent=8.0, seed=5000))):
    ctx = E.Context.on_torch_stream(0)
    with torch.cuda.stream(ctx.torch_stream):
        prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
        opt = E.default_options(); opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
        opt.max_num_iterations = 3
        prob.solve(opt)
        best = 0.0
        for rep in range(3):
            prob.set_params(sc.cams0, sc.pts0)
            opt.max_num_iterations = 20
            ctx.synchronize(); t0 = time.perf_counter()
            s = prob.solve(opt); ctx.synchronize()
            best = max(best, s.num_iterations / (time.perf_counter() - t0))
        prob.set_params(sc.cams0, sc.pts0)
        ctx.set_kernel_timing(True)
        for k in (_lib.K_BA_LINEARIZE, _lib.K_BA_SCHUR, _lib.K_BA_SOLVE): ctx.kernel_time(k)
        s = prob.solve(opt); ctx.synchronize()
        l = ctx.kernel_time(_lib.K_BA_LINEARIZE); sch = ctx.kernel_time(_lib.K_BA_SCHUR); so = ctx.kernel_time(_lib.K_BA_SOLVE)
        ctx.set_kernel_timing(False)
        prob.close()
    print(f"{os.environ.get('ESFM_LIB', 'in-tree'):44s} {name:6s} {best:8.1f} LM it/s  sweep {l[0] / max(l[1], 1) * 1e3:7.1f} us ({l[1]} launches)  schur {sch[0] / max(sch[1], 1) * 1e3:7.1f} us  "
          f"solve {so[0] / max(so[1], 1) * 1e3:7.1f} us  final cost {s.final_cost:.9f}")

"""A BA call of the size an incremental reconstruction makes every ba_frequency frames (a dozen cameras, ~1000 points): where the
milliseconds go -- problem set-up (allocations, uploads, tables), the LM iterations, read-back."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import easysfm_amd as E
from easysfm_amd import synth
ctx = E.Context(0, None)
for n_cam, n_pt, per in ((4, 400, 3), (9, 1200, 4), (11, 1500, 5)):
    sc = synth.ba_scene(n_cam, n_pt, per, seed=5)
    opt = E.default_options()
    E.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, opt, ctx)
    t0 = time.perf_counter()
    for _ in range(10):
        c, p, summ = E.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, opt, ctx)
    one_shot = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
        ctx.synchronize()
    create = (time.perf_counter() - t0) / 10
    prob.set_params(sc.cams0, sc.pts0); ctx.synchronize()
    t0 = time.perf_counter(); s2 = prob.solve(opt); ctx.synchronize(); solve = time.perf_counter() - t0
    t0 = time.perf_counter(); prob.close(); destroy = time.perf_counter() - t0
    print(f"{n_cam} cams {n_pt} pts {len(sc.cam_idx)} obs: one-shot {one_shot*1e3:.2f} ms ({summ.num_iterations} LM iterations); create {create*1e3:.2f} ms, solve {solve*1e3:.2f} ms "
          f"({solve/max(s2.num_iterations,1)*1e3:.3f} ms per iteration), destroy {destroy*1e3:.2f} ms")

"""BA-512: LM it/s of the library in ESFM_LIB (one line)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import easysfm_amd as E
from easysfm_amd import synth
sc = synth.ba_scene(512, 300000, 10, radius=40.0, extent=8.0, seed=5000)
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
print(f"{os.environ.get('ESFM_LIB', 'in-tree'):40s} {best:8.1f} LM it/s  final cost {s.final_cost:.6f}")

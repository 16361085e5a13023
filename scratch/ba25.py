# BA-25 timing: LM iterations/s and per-kernel averages
import sys, time; sys.path.insert(0,'.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
sc = synth.ba_scene(25, 30000, 8, radius=10.0, extent=2.0, seed=4000)
ctx = E.Context(0)
prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
opt = E.default_options(); opt.max_num_iterations = 3; opt.function_tolerance=0; opt.parameter_tolerance=0; opt.gradient_tolerance=0
prob.solve(opt); prob.set_params(sc.cams0, sc.pts0)
opt.max_num_iterations = 50
t=time.time(); s = prob.solve(opt); el=time.time()-t
print('iters', s.num_iterations, 'sec', el, 'it/s', s.num_iterations/el, 'cost', s.initial_cost, s.final_cost)

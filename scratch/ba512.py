import sys, time; sys.path.insert(0,'.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
sc = synth.ba_scene(512, 300000, 10, radius=40.0, extent=8.0, seed=5000)
ctx = E.Context(0)
prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
opt = E.default_options(); opt.max_num_iterations = 2; opt.function_tolerance=0; opt.parameter_tolerance=0
prob.solve(opt); prob.set_params(sc.cams0, sc.pts0)
ctx.set_kernel_timing(True)
opt.max_num_iterations = 8
t=time.time(); s = prob.solve(opt); el=time.time()-t
print('iters', s.num_iterations, 'sec', el, 'it/s', s.num_iterations/el, 'cost', s.initial_cost, s.final_cost)
for k,n in ((_lib.K_BA_LINEARIZE,'linearize'),(_lib.K_BA_SCHUR,'schur'),(_lib.K_BA_SOLVE,'solve')):
    ms,c = ctx.kernel_time(k); print(n, ms/max(c,1), 'ms avg over', c)

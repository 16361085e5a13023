import sys; sys.path.insert(0,'.')
import numpy as np, easysfm_amd as E, oracle
from easysfm_amd import synth
ctx = E.Context(0)
def run(sc, max_iter, **kw):
    opt = E.default_options(); opt.max_num_iterations = max_iter
    ropt = oracle.ba_default_options(); ropt.max_num_iterations = max_iter
    for k,v in kw.items(): setattr(opt,k,v); setattr(ropt,k,v)
    cams,pts,s = E.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, opt, ctx)
    rc,rp,rs = oracle.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ropt)
    print('term', s.termination, rs.termination, 'iters', s.num_iterations, rs.num_iterations)
    for a,b in zip(s.log(), oracle.iterations(rs)):
        print(f"{a.iteration:3d} ok {a.step_is_successful}{b.step_is_successful} cost {a.cost:.12e} {b.cost:.12e} rel {abs(a.cost-b.cost)/max(abs(b.cost),1e-300):.1e} | step {a.step_norm:.6e} {b.step_norm:.6e} | mcc {a.model_cost_change:.6e} {b.model_cost_change:.6e} | rad {a.trust_region_radius:.3e} {b.trust_region_radius:.3e} | g {a.gradient_max_norm:.4e} {b.gradient_max_norm:.4e}")
    print('param max diff cams', np.abs(cams-rc).max(), 'pts', np.abs(pts-rp).max())
print("== 4-50-3"); run(synth.ba_scene(4,50,3,seed=1), 6)
print("== rejected"); run(synth.ba_scene(5,120,4,seed=12,start_noise=(0.08,0.4,0.4)), 12, initial_trust_region_radius=1e7)

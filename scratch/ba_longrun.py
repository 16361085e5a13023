import numpy as np, easysfm_amd as E, oracle
from easysfm_amd import synth
sc = synth.ba_scene(4, 50, 3, seed=1)
opt = E.default_options(); opt.max_num_iterations = 50
ropt = oracle.ba_default_options(); ropt.max_num_iterations = 50
ctx = None
c, p, s = E.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, opt, ctx)
rc, rp, rs = oracle.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ropt)
print("term", s.termination, rs.termination, s.num_iterations, rs.num_iterations)
A = list(s.log()); B = list(oracle.iterations(rs))
for i in range(max(len(A), len(B))):
    a = A[i] if i < len(A) else None; b = B[i] if i < len(B) else None
    f = lambda t: "%2d %.15e ok=%d r=%.3e step=%.3e mcc=%.3e rd=%.3e" % (t.iteration, t.cost, t.step_is_successful, t.trust_region_radius, t.step_norm, t.model_cost_change, t.relative_decrease) if t else "-"
    print(f(a), "|", f(b))

import sys; sys.path.insert(0,'.')
import numpy as np, easysfm_amd as E, oracle
from easysfm_amd import synth
sc = synth.ba_scene(40, 50, 3, seed=3)
e = np.zeros(0, np.int32)
opt = E.default_options(); opt.max_num_iterations = 5
for mode in (None, "sparse", "dense"):
    import os
    os.environ.pop("ESFM_BA_SOLVE", None)
    if mode: os.environ["ESFM_BA_SOLVE"] = mode
    try:
        c, p, s = E.ba_solve(e, e, np.zeros((0, 2), np.float32), sc.K4, sc.cams0, sc.pts0, opt, E.Context(0, None))
        print(mode, "ok", s.num_iterations, s.termination, s.initial_cost, np.array_equal(c, sc.cams0), np.array_equal(p, sc.pts0))
    except Exception as ex:
        print(mode, "FAILED", repr(ex)[:200])
ro = oracle.ba_default_options(); ro.max_num_iterations = 5
rc, rp, rs = oracle.ba_solve(e, e, np.zeros((0, 2), np.float32), sc.K4, sc.cams0, sc.pts0, ro)
print("oracle", rs.num_iterations, rs.termination, rs.initial_cost)
# a few observations on 3 of 40 cameras only (isolated cameras get a plan)
keep = sc.cam_idx < 3
c, p, s = E.ba_solve(sc.cam_idx[keep], sc.pt_idx[keep], sc.uv[keep], sc.K4, sc.cams0, sc.pts0, opt, E.Context(0, None))
rc, rp, rs = oracle.ba_solve(sc.cam_idx[keep], sc.pt_idx[keep], sc.uv[keep], sc.K4, sc.cams0, sc.pts0, ro)
print("3 of 40 cameras observed:", s.num_iterations, rs.num_iterations, [abs(a.cost-b.cost)/max(b.cost,1) for a,b in zip(s.log(), oracle.iterations(rs))])

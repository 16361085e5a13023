import sys; sys.path.insert(0,'.')
import numpy as np, easysfm_amd as E, oracle
from easysfm_amd import synth
oracle.build()
sc = synth.ba_scene(6, 300, 4, seed=2)
uv = sc.uv.copy(); uv[17, 0] = np.nan
opt = E.default_options(); opt.max_num_iterations = 5
ropt = oracle.ba_default_options(); ropt.max_num_iterations = 5
try:
    c, p, s = E.ba_solve(sc.cam_idx, sc.pt_idx, uv, sc.K4, sc.cams0, sc.pts0, opt, E.Context(0, None))
    print('gpu termination', s.termination, s.num_iterations, s.initial_cost, s.final_cost, np.isfinite(c).all())
except Exception as e:
    print('gpu raised', repr(e)[:200])
try:
    rc, rp, rs = oracle.ba_solve(sc.cam_idx, sc.pt_idx, uv, sc.K4, sc.cams0, sc.pts0, ropt)
    print('oracle termination', rs.termination, rs.num_iterations, rs.initial_cost, rs.final_cost)
except Exception as e:
    print('oracle raised', repr(e)[:200])
# a point far behind a camera after a step?  huge initial radius with a bad start
pts0 = sc.pts0.copy(); pts0[5] = [1e30, -1e30, 1e30]
for name, fn, o in (('gpu', lambda: E.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, pts0, opt, E.Context(0, None)), None),
                    ('oracle', lambda: oracle.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, pts0, ropt), None)):
    try:
        c, p, s = fn()
        print(name, 'huge point: termination', s.termination, s.num_iterations, s.initial_cost, s.final_cost)
    except Exception as e:
        print(name, 'raised', repr(e)[:200])

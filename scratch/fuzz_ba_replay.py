import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easysfm_amd as E
import oracle
oracle.build(); oracle.set_num_threads(16)
ctx = E.Context(0)
for f in sys.argv[1:]:
    z = np.load(f)
    for iters in (5, 8):
        opt = E.default_options(); opt.max_num_iterations = iters
        ropt = oracle.ba_default_options(); ropt.max_num_iterations = iters
        cs, ps, ss = E.ba_solve(z["cam"], z["pt"], z["uv"], z["K4"], z["cams0"], z["pts0"], opt, ctx)
        rc, rp, rs = oracle.ba_solve(z["cam"], z["pt"], z["uv"], z["K4"], z["cams0"], z["pts0"], ropt)
        print(f, "iters", iters, "term", ss.termination, rs.termination)
        for a, b in zip(ss.log(), oracle.iterations(rs)):
            print(f"  it {a.iteration}: cost {a.cost:.12e} {b.cost:.12e} rel {abs(a.cost-b.cost)/max(abs(b.cost),1):.2e} | radius {a.trust_region_radius:.9e} {b.trust_region_radius:.9e} rel {abs(a.trust_region_radius-b.trust_region_radius)/b.trust_region_radius:.2e} | valid {a.step_is_valid} {b.step_is_valid} succ {a.step_is_successful} {b.step_is_successful} | step {a.step_norm:.6e} {b.step_norm:.6e}")

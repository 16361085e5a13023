# repeated large solves (the dataflow Cholesky's flags under load): BA-512 and a few mid sizes, many solves, bitwise-equal results
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth
ctx = E.Context(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
t0 = time.time()
for n_cam, n_pt, per, reps, iters in ((512, 300000, 10, 12 * M, 12), (107, 9000, 6, 60 * M, 6), (43, 4000, 6, 80 * M, 5), (30, 2500, 5, 80 * M, 5), (200, 40000, 8, 25 * M, 8)):
    sc = synth.ba_scene(n_cam, n_pt, per, radius=40.0 if n_cam > 150 else 15.0, extent=8.0 if n_cam > 150 else 3.0, seed=5000 + n_cam)
    prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
    opt = E.default_options(); opt.function_tolerance = 0; opt.parameter_tolerance = 0; opt.gradient_tolerance = 0; opt.max_num_iterations = iters
    ref = None; n_iter = 0
    for r in range(reps):
        prob.set_params(sc.cams0, sc.pts0)
        s = prob.solve(opt); ctx.synchronize()
        cams, pts = prob.get_params() if hasattr(prob, 'get_params') else (None, None)
        sig = (s.final_cost, s.num_iterations, s.num_successful_steps, None if cams is None else (cams.tobytes(), pts.tobytes()))
        if ref is None: ref = sig
        assert sig == ref, (n_cam, r, s.final_cost, ref[0])
        n_iter += s.num_iterations
    print(f'{n_cam:4d} cameras: {reps} solves, {n_iter} LM iterations, final cost {ref[0]:.9g}, all bit-identical  ({time.time() - t0:.1f} s)')
    prob.close()

# BA-25 and BA-512: LM iterations/s (timers off) and per-kernel averages (timers on); argv[1] = 25 | 512 | both
import sys, time; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
which = sys.argv[1] if len(sys.argv) > 1 else 'both'
ctx = E.Context(0)
for name, args, iters in (('25', dict(n_cam=25, n_pt=30000, per=8, radius=10.0, extent=2.0, seed=4000), 50),
                          ('512', dict(n_cam=512, n_pt=300000, per=10, radius=40.0, extent=8.0, seed=5000), 12)):
    if which not in (name, 'both'):
        continue
    sc = synth.ba_scene(args['n_cam'], args['n_pt'], args['per'], radius=args['radius'], extent=args['extent'], seed=args['seed'])
    prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
    opt = E.default_options(); opt.function_tolerance = 0; opt.parameter_tolerance = 0; opt.gradient_tolerance = 0
    opt.max_num_iterations = 3; prob.solve(opt)
    best = 0.0
    for rep in range(3):
        prob.set_params(sc.cams0, sc.pts0); opt.max_num_iterations = iters
        ctx.synchronize(); t = time.perf_counter(); s = prob.solve(opt); ctx.synchronize(); el = time.perf_counter() - t
        best = max(best, s.num_iterations / el)
    prob.set_params(sc.cams0, sc.pts0); ctx.set_kernel_timing(True)
    for k in (_lib.K_BA_LINEARIZE, _lib.K_BA_SCHUR, _lib.K_BA_SOLVE): ctx.kernel_time(k)
    s = prob.solve(opt); ctx.synchronize()
    out = {}
    for k, n in ((_lib.K_BA_LINEARIZE, 'linearize'), (_lib.K_BA_SCHUR, 'schur'), (_lib.K_BA_SOLVE, 'solve')):
        ms, c = ctx.kernel_time(k); out[n] = round(ms / max(c, 1) * 1e3, 1)
    ctx.set_kernel_timing(False)
    print(f'BA-{name}: {best:8.1f} LM it/s  ({1e3 / best:.3f} ms/it)  kernels us {out}  cost {s.initial_cost:.4f} -> {s.final_cost:.6f} accepted {s.num_successful_steps}')
    prob.close()

"""BA-512 on one GPU: LM iterations/s and the per-kernel split, dense tiled solve against the structure-aware one (ESFM_BA_SOLVE)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import easysfm_amd as E
from easysfm_amd import synth, _lib

nc, npt, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 300000, 10)))
rad, ext = (40.0, 8.0) if nc >= 400 else (15.0, 3.0)
sc = synth.ba_scene(nc, npt, k, radius=rad, extent=ext, seed=5000)
ctx = E.Context.on_torch_stream(0)
for mode in ("dense", "sparse", "dense", "sparse"):
    os.environ["ESFM_BA_SOLVE"] = mode
    with torch.cuda.stream(ctx.torch_stream):
        prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
        opt = E.default_options(); opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
        opt.max_num_iterations = 3
        prob.solve(opt); prob.set_params(sc.cams0, sc.pts0)
        opt.max_num_iterations = 20
        ctx.synchronize(); t0 = time.perf_counter()
        s = prob.solve(opt); ctx.synchronize()
        el = time.perf_counter() - t0
        prob.set_params(sc.cams0, sc.pts0)
        ctx.set_kernel_timing(True)
        for kid in (_lib.K_BA_LINEARIZE, _lib.K_BA_SCHUR, _lib.K_BA_SOLVE): ctx.kernel_time(kid)
        prob.solve(opt); ctx.synchronize()
        tl = ctx.kernel_time(_lib.K_BA_LINEARIZE); ts = ctx.kernel_time(_lib.K_BA_SCHUR); tc = ctx.kernel_time(_lib.K_BA_SOLVE)
        ctx.set_kernel_timing(False)
        prob.close()
    print(f"{mode:6s}: {s.num_iterations / el:8.1f} LM it/s ({el / s.num_iterations * 1e3:.3f} ms/it), final cost {s.final_cost:.6f}; "
          f"sweep {tl[0] / max(tl[1], 1):.3f} ms, schur {ts[0] / max(ts[1], 1):.3f} ms, solve {tc[0] / max(tc[1], 1):.3f} ms", flush=True)

"""Stage times of ba_schur_mfma_kernel on BA-512 (build: scratch/build_variant_ba.sh schtr -DESFM_SCHUR_TRACE; ESFM_LIB=scratch/variants/libesfm_schtr.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easysfm_amd as E
from easysfm_amd import synth, _lib
sc = synth.ba_scene(512, 300000, 10, radius=40.0, extent=8.0, seed=5000)
ctx = E.Context(0)
prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
opt = E.default_options(); opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
opt.max_num_iterations = 3
prob.solve(opt); prob.set_params(sc.cams0, sc.pts0)
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
lib.esfm_debug_schur_trace(out, 1)
opt.max_num_iterations = 10
ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_BA_SCHUR)
s = prob.solve(opt); ctx.synchronize()
ts = ctx.kernel_time(_lib.K_BA_SCHUR)
lib.esfm_debug_schur_trace(out, 0)
v = [int(x) for x in out]
nb = max(v[3], 1)
print(f"schur {ts[0] / max(ts[1], 1):.3f} ms per call over {ts[1]} calls; batches {v[3]}, workgroups {v[5]}")
print(f"per batch and wave: wait for rows {v[0] / nb * 0.01:.2f} us, W / Y -> LDS {v[1] / nb * 0.01:.2f} us, products {v[2] / nb * 0.01:.2f} us; "
      f"sum + flush per wave and workgroup {v[4] / max(4 * v[5], 1) * 0.01:.2f} us")

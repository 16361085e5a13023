"""Stage times of ba_linearize_kernel on BA-512 (build: scratch/build_variant_ba.sh lintr -DESFM_LIN_TRACE; ESFM_LIB=scratch/variants/libesfm_lintr.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easysfm_amd as E
from easysfm_amd import synth, _lib
sc = synth.ba_scene(512, 300000, 10, radius=40.0, extent=8.0, seed=5000)
ctx = E.Context(0)
prob = E.BAProblem(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ctx)
opt = E.default_options(); opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
opt.max_num_iterations = 4
prob.solve(opt)
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
# continue from the current parameters (no set_params: the first sweeps of a solve are the two-pass form)
lib.esfm_debug_lin_trace(out, 1)
opt.max_num_iterations = 10
ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_BA_LINEARIZE)
s = prob.solve(opt); ctx.synchronize()
ts = ctx.kernel_time(_lib.K_BA_LINEARIZE)
lib.esfm_debug_lin_trace(out, 0)
v = [int(x) for x in out]
nw = max(v[3], 1)
print(f"sweep {ts[0] / max(ts[1], 1):.3f} ms per call over {ts[1]} calls; waves {v[3]} ({v[3] / max(ts[1], 1):.0f} per call), iterations per wave {v[4] / nw:.2f}")
print(f"per wave: prologue {v[0] / nw * 0.01:.2f} us, loop {v[1] / nw * 0.01:.2f} us ({v[1] / max(v[4], 1) * 0.01:.2f} us per iteration), behind the loop {v[2] / nw * 0.01:.2f} us")

import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import easysfm_amd as E
from easysfm_amd import _lib
import ctypes as C
ctx = E.Context(0, None)
ctx.set_kernel_timing(True)
L = E.lib()
for n in (30000, 100000, 300000):
    rng = np.random.default_rng(1)
    P = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
    dP = torch.from_numpy(P).cuda(); out = torch.empty(n, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for it in range(3):
        E._lib.check(L.esfm_sor_mean_distances_dev(ctx.handle, C.c_void_p(dP.data_ptr()), n, 3, 50, C.c_void_p(out.data_ptr())))
    ctx.synchronize()
    ms, cnt = ctx.kernel_time(_lib.K_SOR_KNN)
    steps = (n / 64.0) * n   # wave-steps of 64 candidates
    print(f"n={n} kernel {ms/cnt:.3f} ms  pair evals/s {n*n/(ms/cnt*1e-3):.3e}")

import sys, time; sys.path.insert(0,'.')
import numpy as np, torch, ctypes as C, easysfm_amd as E
from easysfm_amd import synth, _lib
rng = np.random.default_rng(4100)
cloud = np.concatenate([synth.ba_scene(25, 30000, 8, radius=10.0, extent=2.0, seed=4000).pts_gt, rng.uniform(-12, 12, (600, 3))]).astype(np.float32)
n = len(cloud); dev = torch.device('cuda', 0)
d_pts = torch.from_numpy(cloud).to(dev); d_out = torch.empty(n, dtype=torch.float32, device=dev); torch.cuda.synchronize()
ctx = E.Context(0, None); L = E.lib()
call = lambda: _lib.check(L.esfm_sor_mean_distances_dev(ctx.handle, C.c_void_p(d_pts.data_ptr()), n, 3, 50, C.c_void_p(d_out.data_ptr())))
call(); ctx.synchronize(); ctx.set_kernel_timing(True); ctx.kernel_time(_lib.K_SOR_KNN)
for _ in range(10): call()
ctx.synchronize(); ms, k = ctx.kernel_time(_lib.K_SOR_KNN)
print('sor kernel ms', ms / k)

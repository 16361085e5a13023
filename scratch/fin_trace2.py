# stage 2 (threshold filter) of l2_finish_kernel on M-SURF-4k-hard (build with -DESFM_FIN_TRACE2 -DESFM_FIN_NOLOCAL)
import sys; sys.path.insert(0, '.')
import ctypes as C, numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
ctx0 = E.Context(0, None)
imgs = np.load("tests/golden/fountain11_gray.npz")["images"]
pool = np.concatenate([E.surf_detect_and_compute(im, 300.0, None, ctx0)[1] for im in imgs])
pm = E.PairMatcher(E.DescriptorBank(synth.msurf4k_hard_sets(pool), E.ESFM_L2_F32), synth.all_pairs(25))
for _ in range(3): pm.match(0.5)
pm.ctx.synchronize()
out = (C.c_int32 * 16)(); _lib.check(_lib.lib().esfm_match_debug_counters(pm.ctx.handle, out)); c = list(out)
print(c)
print(f"chunks {c[6]}, filter loop {c[4] / max(c[6], 1) / 100:.1f} us per chunk, hits per chunk {c[5] / max(c[6], 1):.0f} (max {c[3]}), hit evaluation + merge {c[7] / max(c[6], 1) / 100:.1f} us per chunk")

# M-SURF-4k-hard, ratio 0.5: a few match() calls, for rocprofv3 passes
import sys; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth
ctx0 = E.Context(0, None)
imgs = np.load("tests/golden/fountain11_gray.npz")["images"]
pool = np.concatenate([E.surf_detect_and_compute(im, 300.0, None, ctx0)[1] for im in imgs])
pm = E.PairMatcher(E.DescriptorBank(synth.msurf4k_hard_sets(pool), E.ESFM_L2_F32), synth.all_pairs(25))
for _ in range(6): pm.match(0.5)
pm.ctx.synchronize()
print("done", int(pm.match(0.5).n_out.sum().item()))

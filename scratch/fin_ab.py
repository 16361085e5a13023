# finish-kernel time on M-SURF-4k (benign) and M-SURF-4k-hard at ratio 0.5 for the library ESFM_LIB points at; checks results against the default library's
import sys, os; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
ctx0 = E.Context(0, None)
imgs = np.load("tests/golden/fountain11_gray.npz")["images"]
pool = np.concatenate([E.surf_detect_and_compute(im, 300.0, None, ctx0)[1] for im in imgs])
pairs = synth.all_pairs(25)
line = []
for name, sets in (("benign", synth.surf_like_sets(25, 4096, pool=16384, seed_base=1000)), ("hard", synth.msurf4k_hard_sets(pool))):
    pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_L2_F32), pairs)
    best = []
    for rep in range(3):
        for _ in range(3): pm.match(0.5)
        pm.ctx.synchronize(); pm.ctx.set_kernel_timing(True); pm.ctx.kernel_time(_lib.K_L2_SECOND)
        for _ in range(20): r = pm.match(0.5)
        pm.ctx.synchronize(); f = pm.ctx.kernel_time(_lib.K_L2_SECOND); pm.ctx.set_kernel_timing(False)
        best.append(f[0] / max(f[1], 1))
    h = r.to_host()
    sig = hash(tuple(int(x[0].sum()) * 31 + int(x[1].sum()) + int(x[2].view(np.uint32).astype(np.uint64).sum()) for x in h))
    line.append(f"{name} finish {min(best):.4f} ms (runs {' '.join(f'{b:.4f}' for b in best)}) sig {sig & 0xffffffff:08x}")
    pm.close()
print(" | ".join(line))

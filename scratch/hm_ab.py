import sys; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
pairs = synth.all_pairs(25)
sets = synth.orb_like_sets(25, 4096, pool=16384, seed_base=3000)
pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_HAMMING), pairs)
out = []
for rep in range(3):
    for _ in range(3): pm.match(0.8)
    pm.ctx.synchronize(); pm.ctx.set_kernel_timing(True); pm.ctx.kernel_time(_lib.K_HAMMING_KNN)
    for _ in range(20): r = pm.match(0.8)
    pm.ctx.synchronize(); k = pm.ctx.kernel_time(_lib.K_HAMMING_KNN); pm.ctx.set_kernel_timing(False)
    out.append(k[0] / max(k[1], 1))
h = r.to_host()
sig = hash(tuple(int(x[0].sum()) * 31 + int(x[1].sum()) + int(x[2].view(np.uint32).astype(np.uint64).sum()) for x in h)) & 0xffffffff
print("  ".join(f"{x:.4f}" for x in out), f"sig {sig:08x}")

# Is l2_knn_bf16_kernel limited by power?  The same launch on descriptors whose values are exactly representable in bf16 (lo halves
# all zero: two of the three MFMA products multiply by zeros) against ordinary f32 descriptors.  Same instruction stream, same
# memory traffic; only the operand bits differ.
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch, easysfm_amd as E
from easysfm_amd import synth, _lib
def run(sets, name):
    pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_L2_F32), synth.all_pairs(25))
    for _ in range(5): pm.match(0.5)
    pm.ctx.synchronize(); pm.ctx.set_kernel_timing(True); pm.ctx.kernel_time(_lib.K_L2_KNN)
    for _ in range(100): pm.match(0.5)
    pm.ctx.synchronize(); ms, n = pm.ctx.kernel_time(_lib.K_L2_KNN)
    print(name, 'knn ms', round(ms / n, 4), 'rescans', pm.stats()[1], flush=True)
sets = synth.surf_like_sets(25, 4096, pool=16384, seed_base=1000)
def to_bf16(x):
    u = x.view(np.uint32); r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)
b = [to_bf16(s.copy()) for s in sets]
for rep in range(2):
    run(sets, 'f32 descriptors          ')
    run(b,    'bf16-exact descriptors   ')

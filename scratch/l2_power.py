# Is the f32-MFMA matcher's time data dependent (matrix-pipe power)?  Same launch on random, low-mantissa and zero descriptors.
import sys, time; sys.path.insert(0, '.')
import numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
pairs = synth.all_pairs(25)
base = synth.surf_like_sets(25, 4096, pool=16384, seed_base=1000)
def run(name, sets):
    bank = E.DescriptorBank(sets, E.ESFM_L2_F32)
    pm = E.PairMatcher(bank, pairs)
    for _ in range(2): pm.match(0.8)
    pm.ctx.synchronize(); pm.ctx.set_kernel_timing(True)
    for _ in range(5): pm.match(0.8)
    pm.ctx.synchronize()
    ms, n = pm.ctx.kernel_time(_lib.K_L2_KNN)
    print(name, 'knn kernel ms', ms / n, 'TFLOP/s', 2 * 300 * 4096 * 4096 * 64 / (ms / n * 1e-3) / 1e12)
run('random', base)
def trunc(a, bits):
    u = a.view(np.uint32) & np.uint32((0xFFFFFFFF << (23 - bits)) & 0xFFFFFFFF)
    return u.view(np.float32)
run('mantissa 8 bits', [trunc(s.copy(), 8) for s in base])
run('mantissa 3 bits', [trunc(s.copy(), 3) for s in base])
run('zeros', [np.zeros_like(s) for s in base])
run('ones', [np.ones_like(s) for s in base])

# per-wave stage times of hamming_fp4_kernel and the shader clock inside its main loop (build with -DESFM_HMX1_TRACE, ESFM_LIB=...)
import sys; sys.path.insert(0, '.')
import ctypes as C, numpy as np, easysfm_amd as E
from easysfm_amd import synth, _lib
pairs = synth.all_pairs(25)
sets = synth.orb_like_sets(25, 4096, pool=16384, seed_base=3000)
pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_HAMMING), pairs)
for _ in range(3): pm.match(0.8)
pm.ctx.synchronize()
L = _lib.lib()
L.esfm_debug_hmx1_trace(None, 1)
pm.ctx.set_kernel_timing(True); pm.ctx.kernel_time(_lib.K_HAMMING_KNN)
reps = 10
for _ in range(reps): pm.match(0.8)
pm.ctx.synchronize()
k = pm.ctx.kernel_time(_lib.K_HAMMING_KNN)
out = (C.c_int32 * 8)(); L.esfm_debug_hmx1_trace(out, 0)
c = list(out); w = max(c[3], 1)
print(f"hamming_fp4_kernel (trace build) {k[0] / max(k[1], 1):.4f} ms per launch; {w // reps} waves per launch")
print('us per block and wave: set-up (operands, first tiles) %.2f  main loop %.2f  tail (exact re-count, stores) %.2f' % (c[0] / w / 100, c[1] / w / 100, c[2] / w / 100))
ghz = (c[4] * 256.0) / (c[1] * 10.0)          # shader cycles per ns
print('shader clock inside the main loop: %.3f GHz (s_memtime cycles per s_memrealtime tick / 10 ns)' % ghz)
mf = 4096 / 32 * 4 * 4          # MFMAs per wave and block: 128 steps of 32 train rows x 4 K-steps x 4 query sets
print('MFMA pipe time of a wave: %d instructions x 8 passes x 4 cycles = %.1f us at that clock; two waves share a SIMD: %.0f %% busy inside the loop'
      % (mf, mf * 32 / ghz / 1e3, 2 * mf * 32 / ghz / 1e3 / (c[1] / w / 100) * 100))

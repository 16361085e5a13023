"""Per-step time of the headline workload right after a cold start: 5 warm-up steps, then 60 steps timed in groups of 5 (one sync per group)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, easysfm_amd as E
from easysfm_amd import synth
sets = synth.surf_like_sets(25, 4096, pool=16384, seed_base=1000)
pairs = E.shard_pair_list(25, np.full(25, 4096, np.int32), 0, 1)
bank = E.DescriptorBank(sets, E.ESFM_L2_F32, device="cuda:0")
pm = E.PairMatcher(bank, pairs)
for _ in range(5): pm.match(0.5)
pm.ctx.synchronize()
out = []
for g in range(24):
    t0 = time.perf_counter()
    for _ in range(5): pm.match(0.5)
    pm.ctx.synchronize()
    out.append((time.perf_counter() - t0) / 5 * 1e3)
print("ms per step, groups of 5 after 5 warm-up steps:", " ".join(f"{x:.3f}" for x in out))

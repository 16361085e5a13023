# run-to-run consistency of the full-size matchers (races in the asynchronous row transfers would show as differences)
import sys; sys.path.insert(0, '.')
import numpy as np, torch, easysfm_amd as E
from easysfm_amd import synth
for name, sets, metric in (("surf", synth.surf_like_sets(25, 4096, pool=16384, seed_base=1000), E.ESFM_L2_F32),
                           ("orb", synth.orb_like_sets(25, 4096, pool=16384, seed_base=3000), E.ESFM_HAMMING)):
    pm = E.PairMatcher(E.DescriptorBank(sets, metric), synth.all_pairs(25))
    idx, dist = pm.knn2(); pm.ctx.synchronize()
    ref_i, ref_d = idx.clone(), dist.clone()
    nd = 0
    for r in range(60):
        idx, dist = pm.knn2(); pm.ctx.synchronize()
        nd += int((idx != ref_i).any().item()) + int((dist.view(torch.int32) != ref_d.view(torch.int32)).any().item())
    print(name, '60 repeats, differing runs:', nd, flush=True)

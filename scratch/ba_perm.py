import numpy as np, easysfm_amd as E, oracle
from easysfm_amd import synth
for (nc, npt, k, seed, it) in [(5, 150, 4, 10, 8), (25, 2000, 8, 3, 8)]:
    sc = synth.ba_scene(nc, npt, k, seed=seed)
    opt = E.default_options(); opt.max_num_iterations = it
    ropt = oracle.ba_default_options(); ropt.max_num_iterations = it
    c, p, s = E.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, opt)
    perm = np.random.default_rng(0).permutation(sc.n_obs)
    c2, p2, s2 = E.ba_solve(sc.cam_idx[perm], sc.pt_idx[perm], sc.uv[perm], sc.K4, sc.cams0, sc.pts0, opt)
    rc, rp, rs = oracle.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, ropt)
    rc2, rp2, rs2 = oracle.ba_solve(sc.cam_idx[perm], sc.pt_idx[perm], sc.uv[perm], sc.K4, sc.cams0, sc.pts0, ropt)
    print(nc, "gpu perm diff cams %.2e pts %.2e | gpu-oracle cams %.2e pts %.2e | oracle perm diff cams %.2e pts %.2e" % (
        np.abs(c - c2).max(), np.abs(p - p2).max(), np.abs(c - rc).max(), np.abs(p - rp).max(), np.abs(rc - rc2).max(), np.abs(rp - rp2).max()))
    print("   cost rel diff gpu-perm %.2e gpu-oracle %.2e" % (abs(s.final_cost - s2.final_cost) / s.final_cost, abs(s.final_cost - rs.final_cost) / s.final_cost))

"""Randomised stress of the batched matcher against the oracle: ragged set sizes (incl. 0, 1, tile edges), clustered / duplicated /
scaled rows, random ratios, L2 (64 and 128 floats) and 256-bit Hamming; match lists and 2-NN tables, bit for bit.
usage: python tests/stress_match.py [--seconds S | --cases N] [--seed K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easysfm_amd as E
from easysfm_amd import synth
import oracle

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=None, help="run until this much time has passed")
ap.add_argument("--cases", type=int, default=None, help="... or for exactly this many cases (deterministic in --seed)")
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
budget = args.seconds if args.seconds is not None else (1e9 if args.cases is not None else 120.0)
max_cases = args.cases if args.cases is not None else 1 << 60
seed = args.seed
rng = np.random.default_rng(seed)
oracle.build()
lib = oracle
lib.set_num_threads(os.cpu_count() or 1)
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
EDGE = [0, 1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2049, 4095, 4096, 4097]

def l2_sets(n_sets, dim):
    style = rng.integers(0, 5)
    pool = rng.standard_normal((int(rng.integers(8, 3000)), dim))
    if style == 1: pool = pool[: max(2, len(pool) // 50)]                      # few distinct rows: many duplicates / near ties
    sets = []
    for _ in range(n_sets):
        n = int(rng.choice(EDGE)) if rng.random() < 0.5 else int(rng.integers(0, 3000))
        ids = rng.integers(0, len(pool), n)
        noise = [0.0, 1e-7, 1e-3, 0.02, 0.3][int(rng.integers(0, 5))]
        x = pool[ids] + noise * rng.standard_normal((n, dim))
        if style != 2: x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12)
        if style == 3: x *= 10.0 ** rng.uniform(-3, 3)                            # un-normalised magnitudes
        if style == 4 and n: x[rng.integers(0, n, max(1, n // 20))] = 0.0         # zero rows
        sets.append(x.astype(np.float32))
    return sets

def ham_sets(n_sets):
    pool = rng.integers(0, 256, (int(rng.integers(4, 3000)), 32), dtype=np.uint8)
    sets = []
    for _ in range(n_sets):
        n = int(rng.choice(EDGE)) if rng.random() < 0.5 else int(rng.integers(0, 3000))
        x = pool[rng.integers(0, len(pool), n)].copy()
        flip = [0.0, 0.004, 0.05, 0.3][int(rng.integers(0, 4))]
        if flip and n: x ^= np.packbits(rng.random((n, 256)) < flip, axis=1)
        sets.append(x)
    return sets

t_end = time.time() + budget
n_cases = n_pairs = n_queries = 0
while time.time() < t_end and n_cases < max_cases:
    kind = int(rng.integers(0, 3))
    n_sets = int(rng.integers(2, 7))
    ratio = float([0.5, 0.8, 0.6, 0.95, 1.0, 0.3][int(rng.integers(0, 6))])
    pairs = [(i, j) for i in range(n_sets) for j in range(n_sets) if i != j and rng.random() < 0.6] or [(1, 0)]
    if kind == 2:
        sets = ham_sets(n_sets)
        pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_HAMMING), pairs)
        ref_match = lambda a, b: lib.match_hamming(a, b, ratio)
        ref_knn = lib.knn2_hamming
    else:
        sets = l2_sets(n_sets, 64 if kind == 0 else 128)
        pm = E.PairMatcher(E.DescriptorBank(sets, E.ESFM_L2_F32), pairs)
        ref_match = lambda a, b: lib.match_l2(a, b, ratio)
        ref_knn = lib.knn2_l2
    res = pm.match(ratio).to_host()
    idx, dist = pm.knn2(); pm.ctx.synchronize()
    idx = idx.cpu().numpy(); dist = dist.cpu().numpy()
    off = 0
    for (i, j), (qi, ti, d) in zip(pairs, res):
        nq = len(sets[i])
        if nq and len(sets[j]) >= 2:
            rq, rt, rd = ref_match(sets[i], sets[j])
            ok = np.array_equal(qi, rq) and np.array_equal(ti, rt) and (np.array_equal(bits(d), bits(rd)) if kind != 2 else np.array_equal(np.asarray(d, np.float32), np.asarray(rd, np.float32)))
            if not ok:
                print("MATCH MISMATCH", dict(seed=seed, case=n_cases, kind=kind, pair=(i, j), nq=nq, nt=len(sets[j]), ratio=ratio, got=len(qi), want=len(rq)), flush=True)
                os.makedirs("gpurun_out", exist_ok=True); np.savez(f"gpurun_out/stress_match_fail_{seed}_{n_cases}.npz", q=sets[i], t=sets[j], ratio=ratio, kind=kind)
                sys.exit(1)
        else:
            assert len(qi) == 0, ("expected nothing", i, j, nq, len(sets[j]), len(qi))
        if nq:
            ridx, rdist = ref_knn(sets[i], sets[j])
            gi, gd = idx[off:off + nq], dist[off:off + nq]
            same = np.array_equal(gi, ridx) and (np.array_equal(bits(gd), bits(rdist)) if kind != 2 else np.array_equal(np.asarray(gd, np.float32), np.asarray(rdist, np.float32)))
            if not same:
                bad = np.nonzero(np.any(gi != ridx, axis=1))[0][:5]
                print("KNN MISMATCH", dict(seed=seed, case=n_cases, kind=kind, pair=(i, j), nq=nq, nt=len(sets[j]), rows=bad.tolist(), got=gi[bad].tolist(), want=ridx[bad].tolist()), flush=True)
                os.makedirs("gpurun_out", exist_ok=True); np.savez(f"gpurun_out/stress_match_fail_{seed}_{n_cases}.npz", q=sets[i], t=sets[j], ratio=ratio, kind=kind)
                sys.exit(1)
        off += nq
        n_pairs += 1; n_queries += nq
    pm.close()
    n_cases += 1
print(f"stress_match seed {seed}: {n_cases} cases, {n_pairs} pairs, {n_queries} queries, all equal to the oracle")

"""How far the reference's APPROXIMATE SURF matcher sits from the exact one this repo implements (VERDICT r04 "missing" 7).

`matchFeaturesSURF` uses `cv::FlannBasedMatcher` with its defaults (cpp_code/src/feature_matching.cpp:120,125): four randomised kd-trees
over the train descriptors, built again for every pair, searched with 32 leaf checks.  Its output is not a function of its inputs (the
trees are seeded by rand()), so "bit-exact" is only definable against exact brute force -- what the Python prototype does
(python_code/feature_match.py:33-34) and what the C ABI implements.  Nothing said how much the two differ.  This file restates FLANN's
KDTreeIndex [upstream: flann/algorithms/kdtree_index.h -- mean split on one of the five highest-variance dimensions of a 100-point
sample, one point per leaf, best-bin-first search over all trees through one branch heap, `checks` leaves in total] in numpy / pure
Python and measures, on the metric's generator and on the reference's own fountain images:
  * recall of the nearest and of the second-nearest neighbour,
  * how the ratio-test match list (`d0 < 0.5 d1`, :133) differs: exact matches FLANN loses, and matches FLANN emits that exact search
    rejects (a missed second neighbour makes d1 too large and the ratio test too easy).
CPU only (the exact side is the oracle).  The numbers are quoted in DESIGN.md section 2; the asserts are loose sanity bounds."""
import heapq
import os

import numpy as np
import pytest

from easysfm_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


class KDForest:
    """FLANN KDTreeIndex(trees) over `data` (float32 [n, dim]) [upstream semantics restated; the random choices are this file's]."""
    SAMPLE_MEAN, RAND_DIM = 100, 5

    def __init__(self, data, trees=4, seed=0):
        self.data = np.ascontiguousarray(data, np.float32)
        self.rng = np.random.default_rng(seed)
        self.nodes = []                     # (dim, cut, left, right) or (-1, point, -1, -1)
        self.roots = []
        for _ in range(trees):
            idx = self.rng.permutation(len(self.data))
            self.roots.append(self._divide(idx))

    def _divide(self, idx):
        if len(idx) == 1:
            self.nodes.append((-1, int(idx[0]), -1, -1))
            return len(self.nodes) - 1
        sample = self.data[idx[:self.SAMPLE_MEAN]]
        mean = sample.mean(axis=0)
        var = ((sample - mean) ** 2).sum(axis=0)
        top = np.argsort(-var, kind="stable")[:self.RAND_DIM]
        dim = int(top[self.rng.integers(0, self.RAND_DIM)])
        cut = float(mean[dim])
        v = self.data[idx, dim]
        left, right = idx[v < cut], idx[v >= cut]
        if len(left) == 0 or len(right) == 0:          # (FLANN: split in the middle when the plane separates nothing)
            left, right = idx[:len(idx) // 2], idx[len(idx) // 2:]
        me = len(self.nodes)
        self.nodes.append(None)
        l = self._divide(left); r = self._divide(right)
        self.nodes[me] = (dim, cut, l, r)
        return me

    def knn2(self, q, checks=32):
        """Approximate two nearest neighbours of q: (idx0, idx1, d0, d1) with squared distances; -1 when fewer were found."""
        data, nodes = self.data, self.nodes
        heap, checked, best = [], set(), []           # best: up to two (dist, idx), sorted
        count = [0]

        def descend(node, mindist):
            while True:
                dim, cut, l, r = nodes[node]
                if dim < 0:
                    p = cut
                    if p in checked or (count[0] >= checks and len(best) == 2):
                        return
                    checked.add(p); count[0] += 1
                    d = float(((data[p] - q) ** 2).sum())
                    best.append((d, p)); best.sort(); del best[2:]
                    return
                diff = float(q[dim]) - cut
                near, far = (l, r) if diff < 0 else (r, l)
                nd = mindist + diff * diff
                if len(best) < 2 or nd < best[-1][0]:
                    heapq.heappush(heap, (nd, far))
                node = near

        for root in self.roots:
            descend(root, 0.0)
        while heap and (count[0] < checks or len(best) < 2):
            nd, node = heapq.heappop(heap)
            if len(best) == 2 and nd >= best[-1][0]:
                continue
            descend(node, nd)
        while len(best) < 2:
            best.append((np.inf, -1))
        return best[0][1], best[1][1], best[0][0], best[1][0]


def _compare(q, t, oracle_lib, ratio=0.5, seed=0):
    forest = KDForest(t, trees=4, seed=seed)
    ridx, rdist = oracle_lib.knn2_l2(q, t)
    hit0 = hit1 = 0
    flann_list, exact_list = {}, {}
    for i in range(len(q)):
        i0, i1, d0, d1 = forest.knn2(q[i], checks=32)
        hit0 += i0 == ridx[i, 0]
        hit1 += i1 == ridx[i, 1]
        if i1 >= 0 and np.sqrt(np.float32(d0)) < ratio * np.sqrt(np.float32(d1)):
            flann_list[i] = i0
        if ridx[i, 1] >= 0 and float(rdist[i, 0]) < ratio * float(rdist[i, 1]):
            exact_list[i] = int(ridx[i, 0])
    both = sum(1 for k, v in exact_list.items() if flann_list.get(k) == v)
    lost = len(exact_list) - both
    extra = sum(1 for k in flann_list if k not in exact_list)
    wrong = sum(1 for k, v in flann_list.items() if k in exact_list and exact_list[k] != v)
    return dict(n=len(q), recall0=hit0 / len(q), recall1=hit1 / len(q), exact=len(exact_list), flann=len(flann_list), both=both, lost=lost,
                extra=extra, wrong=wrong)


def test_flann_defaults_on_the_metric_generator(oracle_lib):
    s = synth.surf_like_sets(2, 2048, pool=4096, seed_base=1000)
    r = _compare(s[1], s[0], oracle_lib)
    print(f"\nsurf_like_sets 2048 x 2048: FLANN(4 trees, 32 checks) finds the nearest row for {100 * r['recall0']:.1f} % of the queries, the second nearest for "
          f"{100 * r['recall1']:.1f} %; ratio-0.5 lists: exact {r['exact']}, FLANN {r['flann']}, common {r['both']}, lost {r['lost']}, FLANN-only {r['extra']}, other row {r['wrong']}")
    assert r["exact"] > 100 and r["both"] >= 0.5 * r["exact"]            # planted tracks are far nearer than anything else: mostly found
    assert r["recall1"] < 0.9                                              # ... but the isotropic second neighbour is a coin toss at 32 checks


def test_flann_defaults_on_the_fountain_images(oracle_lib):
    imgs = np.load(os.path.join(GOLD, "fountain11_gray.npz"))["images"]
    d = [oracle_lib.surf(imgs[k], 300.0)[1] for k in (4, 5)]
    q, t = d[1][:1200], d[0]
    r = _compare(q, t, oracle_lib)
    print(f"\nfountain images 5 -> 4, SURF-300 ({len(q)} x {len(t)}): FLANN(4 trees, 32 checks) finds the nearest row for {100 * r['recall0']:.1f} % of the queries, "
          f"the second nearest for {100 * r['recall1']:.1f} %; ratio-0.5 lists: exact {r['exact']}, FLANN {r['flann']}, common {r['both']}, lost {r['lost']}, "
          f"FLANN-only {r['extra']}, other row {r['wrong']}")
    assert r["exact"] > 50 and r["both"] >= 0.5 * r["exact"]
    assert r["flann"] >= r["both"]

"""Host-side mirror of the reference's matching interface over the C ABI.

``FeatureMatching.matchFeaturesSURF / matchFeaturesORB`` keep the names, argument meaning and
append-to-``matches`` behaviour of cpp_code/include/feature_matching.h:17-21 (bodies:
cpp_code/src/feature_matching.cpp:71-97, :115-142).  ``match_all_pairs`` is the batched pair loop of
cpp_code/test/sfm.cpp:140-161 with descriptors resident in HBM.  All compute happens in
libesfm_hip.so; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import ESFM_HAMMING, ESFM_L2_F32, Context, check, default_context, lib
from .types import DMatch, Frame


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def _as_desc(a, dtype) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype)
    if a.ndim != 2:
        raise ValueError("descriptors must be a 2-D array [rows, width]")
    return a


def knn_match_l2(q, t, ctx: Optional[Context] = None) -> Tuple[np.ndarray, np.ndarray]:
    """knnMatch(q, t, out, 2) under NORM_L2 (exact brute force).  Returns (idx[nq,2], dist[nq,2])."""
    ctx = ctx or default_context()
    q = _as_desc(q, np.float32); t = _as_desc(t, np.float32)
    if q.shape[1] != t.shape[1]:
        raise ValueError("descriptor widths differ")
    nq = q.shape[0]
    idx = np.full((nq, 2), -1, np.int32); dist = np.zeros((nq, 2), np.float32)
    check(lib().esfm_knn2_l2_f32(ctx.handle, _ptr(q), nq, _ptr(t), t.shape[0], q.shape[1], _ptr(idx), _ptr(dist)))
    return idx, dist


def knn_match_hamming(q, t, ctx: Optional[Context] = None) -> Tuple[np.ndarray, np.ndarray]:
    """knnMatch(q, t, out, 2) for "BruteForce-Hamming".  Returns (idx[nq,2], dist[nq,2])."""
    ctx = ctx or default_context()
    q = _as_desc(q, np.uint8); t = _as_desc(t, np.uint8)
    if q.shape[1] != t.shape[1]:
        raise ValueError("descriptor widths differ")
    nq = q.shape[0]
    idx = np.full((nq, 2), -1, np.int32); dist = np.zeros((nq, 2), np.float32)
    check(lib().esfm_knn2_hamming(ctx.handle, _ptr(q), nq, _ptr(t), t.shape[0], q.shape[1], _ptr(idx), _ptr(dist)))
    return idx, dist


def _match(fn, q, t, ratio, ctx):
    nq = q.shape[0]
    qi = np.empty(max(nq, 1), np.int32); ti = np.empty(max(nq, 1), np.int32); d = np.empty(max(nq, 1), np.float32)
    n = C.c_int32(0)
    check(fn(ctx.handle, _ptr(q), nq, _ptr(t), t.shape[0], q.shape[1], float(ratio), _ptr(qi), _ptr(ti), _ptr(d), C.byref(n)))
    return qi[:n.value].copy(), ti[:n.value].copy(), d[:n.value].copy()


def match_l2(q, t, ratio: float = 0.5, ctx: Optional[Context] = None):
    """2-NN + Lowe ratio for float descriptors; returns (queryIdx, trainIdx, distance) arrays."""
    ctx = ctx or default_context()
    return _match(lib().esfm_match_l2_f32, _as_desc(q, np.float32), _as_desc(t, np.float32), ratio, ctx)


def match_hamming(q, t, ratio: float = 0.8, ctx: Optional[Context] = None):
    ctx = ctx or default_context()
    return _match(lib().esfm_match_hamming, _as_desc(q, np.uint8), _as_desc(t, np.uint8), ratio, ctx)


def match_pairs_host(sets: Sequence[np.ndarray], pairs, ratio: float, metric: int = ESFM_L2_F32, ctx: Optional[Context] = None):
    """esfm_match_pairs: the batched pair loop through HOST pointers (upload once, one launch sequence, one read-back) -- what the
    C++ driver calls.  Returns [(queryIdx, trainIdx, distance)] per pair."""
    ctx = ctx or default_context()
    dt = np.float32 if metric == ESFM_L2_F32 else np.uint8
    sets = [_as_desc(s_, dt) for s_ in sets]
    width = sets[0].shape[1]
    off = np.zeros(len(sets) + 1, np.int32)
    np.cumsum([s_.shape[0] for s_ in sets], out=off[1:])
    bank = np.ascontiguousarray(np.concatenate(sets, axis=0)) if off[-1] else np.zeros((1, width), dt)
    pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    total = int(sum(sets[i].shape[0] for i, _ in pairs))
    qi = np.zeros(max(total, 1), np.int32); ti = np.zeros(max(total, 1), np.int32); d = np.zeros(max(total, 1), np.float32)
    n_out = np.zeros(max(len(pairs), 1), np.int32); out_off = np.zeros(len(pairs) + 1, np.int64)
    check(lib().esfm_match_pairs(ctx.handle, metric, _ptr(bank), _ptr(off), len(sets), width, _ptr(pairs), len(pairs), float(ratio),
                                 _ptr(qi), _ptr(ti), _ptr(d), _ptr(n_out), _ptr(out_off)))
    return [(qi[o:o + k].copy(), ti[o:o + k].copy(), d[o:o + k].copy()) for o, k in zip(out_off[:-1], n_out[:len(pairs)])]


class FeatureMatching:
    """Mirror of p3dv::FeatureMatching's matching members (feature_matching.h:17-21)."""

    def __init__(self, ctx: Optional[Context] = None):
        self._ctx = ctx

    @property
    def ctx(self) -> Context:
        return self._ctx or default_context()

    def matchFeaturesORB(self, cur_frame_1: Frame, cur_frame_2: Frame, matches: List[DMatch],
                         ratio_thre: float = 0.8, show: bool = False) -> bool:
        """feature_matching.cpp:71-97.  Query = cur_frame_1, train = cur_frame_2; survivors are
        APPENDED to `matches` (the reference never clears it, :90)."""
        qi, ti, d = match_hamming(cur_frame_1.descriptors, cur_frame_2.descriptors, ratio_thre, self.ctx)
        matches.extend(DMatch(int(a), int(b), float(c)) for a, b, c in zip(qi, ti, d))
        return True

    def matchFeaturesSURF(self, cur_frame_1: Frame, cur_frame_2: Frame, matches: List[DMatch],
                          ratio_thre: float = 0.5, show: bool = False) -> bool:
        """feature_matching.cpp:115-142 with the exact brute-force matcher the Python prototype
        uses (feature_match.py:33-34) instead of the approximate FlannBasedMatcher (:120)."""
        qi, ti, d = match_l2(cur_frame_1.descriptors, cur_frame_2.descriptors, ratio_thre, self.ctx)
        matches.extend(DMatch(int(a), int(b), float(c)) for a, b, c in zip(qi, ti, d))
        return True

    def detectFeaturesSURF(self, cur_frame: Frame, minHessian: int = 400, show: bool = False) -> bool:
        """feature_matching.cpp:43-69 (features.detectFeaturesSURF)."""
        from .features import detectFeaturesSURF
        return detectFeaturesSURF(cur_frame, minHessian, show, self._ctx)

    def detectFeaturesORB(self, cur_frame: Frame, max_num: int = 5000, show: bool = False) -> bool:
        """feature_matching.cpp:14-41 (features.detectFeaturesORB)."""
        from .features import detectFeaturesORB
        return detectFeaturesORB(cur_frame, max_num, show, self._ctx)

    # ---- frame selection (feature_matching.cpp:160-268): integer logic on the track matrix, host side ----
    def findInitializeFramePair(self, feature_track_matrix, frames, img_match_graph, min_track_num_init: int = 100,
                                max_depth_baseline_ratio_init: float = 50.0):
        """feature_matching.cpp:160-229 (defaults feature_matching.h:26).  feature_track_matrix: [n_frames, n_unique] bool (frame sees track);
        img_match_graph[i][j].appro_depth (or a 2-D array of depths): depth / baseline of pair (i, j < i).
        Returns (found, initialization_frame_1, initialization_frame_2, depth_init).  The pair maximising the sum, over
        the tracks both frames see, of the number of frames that see the track; ties go to the LATER pair in (i, j < i)
        order (`>=` at :203); pairs whose depth / baseline exceeds the limit are skipped (:193-194)."""
        T = np.asarray(feature_track_matrix, bool)
        n_frames = len(frames)
        T = T[:n_frames]
        weight = T.sum(axis=0).astype(np.int64)                    # point_track_frame_num (:172-180)
        score = (T.astype(np.int64) * weight[None, :]) @ T.T.astype(np.int64)

        def depth(i, j):
            e = img_match_graph[i][j]
            return float(getattr(e, "appro_depth", e))

        best, f1, f2, d_init = int(min_track_num_init), 0, 0, None
        for i in range(n_frames):
            for j in range(i):
                d = depth(i, j)
                if d > max_depth_baseline_ratio_init:
                    continue
                if score[i, j] >= best:
                    best, f1, f2, d_init = int(score[i, j]), i, j, d
        if f1 == f2:
            print("Failed to find proper frame pair for initialization. Use default frame [1] and frame [0]")
            return False, 1, 0, None                                # depth_init is left unset by the reference (:217-223)
        return True, f1, f2, d_init

    def findNextFrame(self, feature_track_matrix, frames_to_process, unique_3d_point_ids, next_frame: int = -1) -> int:
        """feature_matching.cpp:231-268: among the frames still to process, the one seeing most of the already
        triangulated tracks; the FIRST such frame on ties (strict `>` at :252); `next_frame` is returned unchanged when
        no candidate sees any (the reference leaves its output argument untouched)."""
        T = np.asarray(feature_track_matrix, bool)
        ids = np.asarray(unique_3d_point_ids, np.int64)
        best = 0
        for i, todo in enumerate(frames_to_process):
            if not todo:
                continue
            c = int(T[i, ids].sum()) if len(ids) else 0
            if c > best:
                best, next_frame = c, i
        return next_frame


# ------------------------------------------------------------------------------------------------
@dataclass
class PairMatches:
    """Result of a batched call: pair p owns out slice [offset[p], offset[p] + n_out[p])."""
    pairs: np.ndarray        # int32 [n_pairs, 2] (query set, train set)
    offset: np.ndarray       # int64 [n_pairs + 1]
    n_out: "object"          # device int32 [n_pairs]   (torch tensor)
    query_idx: "object"      # device int32 [sum nq]
    train_idx: "object"      # device int32
    distance: "object"       # device float32
    ctx: "object" = None

    def to_host(self) -> List[Tuple[np.ndarray, np.ndarray, np.ndarray]]:
        if self.ctx is not None:
            self.ctx.synchronize()
        n = self.n_out.cpu().numpy()
        q = self.query_idx.cpu().numpy(); t = self.train_idx.cpu().numpy(); d = self.distance.cpu().numpy()
        out = []
        for p in range(len(self.pairs)):
            o = int(self.offset[p]); k = int(n[p])
            out.append((q[o:o + k].copy(), t[o:o + k].copy(), d[o:o + k].copy()))
        return out


class DescriptorBank:
    """All descriptor sets of a reconstruction, concatenated in one HBM buffer (torch owns the
    memory; the kernels see a raw device pointer)."""

    def __init__(self, sets: Sequence[np.ndarray], metric: int, device: str = "cuda:0"):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no MI355X visible: DescriptorBank needs a GPU (easysfm_amd has no CPU fallback)")
        self.metric = metric
        dt = np.float32 if metric == ESFM_L2_F32 else np.uint8
        sets = [_as_desc(s, dt) for s in sets]
        widths = {s.shape[1] for s in sets}
        if len(widths) != 1:
            raise ValueError("all descriptor sets must have the same width")
        self.width = widths.pop()
        self.rows = np.array([s.shape[0] for s in sets], np.int32)
        self.row_offset = np.zeros(len(sets) + 1, np.int32)
        np.cumsum(self.rows, out=self.row_offset[1:])
        host = np.concatenate(sets, axis=0) if sets else np.zeros((0, self.width), dt)
        self.device = torch.device(device)
        self.data = torch.from_numpy(host).to(self.device)
        self._matchers = []          # live PairMatchers (weak): update() re-prepares them

    @property
    def n_sets(self) -> int:
        return len(self.rows)

    def update(self, sets: Sequence[np.ndarray]) -> None:
        """Replace the rows IN PLACE (same set sizes: the buffer and every PairMatcher's output slices stay as they are).  The
        operand images a PairMatcher derived from the old rows are stale after this: every matcher of this bank is re-prepared
        (esfm.h: a prepared buffer must be prepared again before a match call relies on it)."""
        import torch
        dt = np.float32 if self.metric == ESFM_L2_F32 else np.uint8
        sets = [_as_desc(s, dt) for s in sets]
        if [s.shape[0] for s in sets] != list(self.rows) or any(s.shape[1] != self.width for s in sets):
            raise ValueError("update() keeps the bank's shape: same number of sets, rows per set and width")
        for m in list(self._matchers):
            m.release()
        if len(sets) and int(self.row_offset[-1]):
            self.data.copy_(torch.from_numpy(np.concatenate(sets, axis=0)))
            torch.cuda.synchronize(self.device)
        for m in list(self._matchers):
            m.prepare()


class _WeakCall:
    """A weak reference that forwards attribute access to its referent (DescriptorBank keeps its matchers without owning them)."""

    def __init__(self, ref):
        self._ref = ref

    def __call__(self):
        return self._ref()

    def __getattr__(self, name):
        obj = self._ref()
        if obj is None:
            return lambda *a, **k: None
        return getattr(obj, name)


class PairMatcher:
    """Batched pair loop (sfm.cpp:140-161): one call matches a whole pair list on one GPU.
    Output buffers are allocated once and reused across calls."""

    def __init__(self, bank: DescriptorBank, pairs: np.ndarray, ctx: Optional[Context] = None):
        import torch
        self.bank = bank
        self.pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        self.torch = torch
        dev_index = bank.device.index or 0
        if ctx is None:
            ctx = Context.on_torch_stream(dev_index)
        self.ctx = ctx
        total = int(bank.rows[self.pairs[:, 0]].sum()) if len(self.pairs) else 0
        self.total_queries = total
        dev = bank.device
        self.query_idx = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        self.train_idx = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        self.distance = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
        self.n_out = torch.empty(max(len(self.pairs), 1), dtype=torch.int32, device=dev)
        self.offset = np.zeros(len(self.pairs) + 1, np.int64)
        self.knn_idx = None
        self.knn_dist = None
        torch.cuda.synchronize(dev)   # the bank upload ran on torch's stream; kernels run on ctx's
        import weakref
        bank._matchers = [m for m in bank._matchers if m() is not None]
        self._ref = weakref.ref(self)
        bank._matchers.append(_WeakCall(self._ref))
        self.prepare()

    def prepare(self) -> None:
        """Part of the upload: the per-row operands the matcher derives from the resident descriptors (esfm_match_prepare_dev) --
        computed once here instead of at the head of every match() / knn2() call.  ONE prepared buffer per context (esfm.h): a
        second PairMatcher on the same Context takes the prepared state over, and the first one's calls re-derive the operands
        every time (correct, slower) until its prepare() is called again.  The bank's rows must not change while prepared:
        DescriptorBank.update() is the way to rewrite them."""
        b = self.bank
        check(lib().esfm_match_prepare_dev(self.ctx.handle, b.metric, C.c_void_p(b.data.data_ptr()), int(b.row_offset[-1]), b.width))

    def release(self) -> None:
        """esfm_match_release_prepared_buffer: to be called before the bank's buffer is freed or rewritten (close() / garbage collection /
        DescriptorBank.update() do) -- a later allocation at the same address must not inherit this one's operand images.  Releases
        only if THIS bank's buffer is the one prepared on the context: an older matcher that is garbage-collected must not take the
        prepared operands from the live one on the same Context (its calls would silently re-derive them every time)."""
        try:
            h = self.ctx.handle
        except RuntimeError:
            return
        check(lib().esfm_match_release_prepared_buffer(h, C.c_void_p(self.bank.data.data_ptr())))

    def close(self) -> None:
        self.release()
        self.bank._matchers = [m for m in self.bank._matchers if m() is not None and m() is not self]

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def set_prepared_check(self, enable: bool) -> None:
        """esfm_ctx_set_prepared_check: fingerprint the prepared buffer and verify it on every call (a debugging aid)."""
        check(lib().esfm_ctx_set_prepared_check(self.ctx.handle, 1 if enable else 0))

    def match(self, ratio: float) -> PairMatches:
        """Enqueue the whole pair list; does not synchronise."""
        b = self.bank
        check(lib().esfm_match_pairs_dev(
            self.ctx.handle, b.metric, C.c_void_p(b.data.data_ptr()), _ptr(b.row_offset), b.n_sets, b.width,
            _ptr(self.pairs), len(self.pairs), float(ratio),
            C.c_void_p(self.query_idx.data_ptr()), C.c_void_p(self.train_idx.data_ptr()),
            C.c_void_p(self.distance.data_ptr()), C.c_void_p(self.n_out.data_ptr()), _ptr(self.offset)))
        return PairMatches(self.pairs, self.offset, self.n_out, self.query_idx, self.train_idx, self.distance, self.ctx)

    def knn2(self):
        """Raw 2-NN table of every pair: (idx[sum nq, 2], dist[sum nq, 2]) device tensors."""
        torch = self.torch
        b = self.bank
        if self.knn_idx is None:
            self.knn_idx = torch.empty((max(self.total_queries, 1), 2), dtype=torch.int32, device=b.device)
            self.knn_dist = torch.empty((max(self.total_queries, 1), 2), dtype=torch.float32, device=b.device)
        check(lib().esfm_knn2_pairs_dev(
            self.ctx.handle, b.metric, C.c_void_p(b.data.data_ptr()), _ptr(b.row_offset), b.n_sets, b.width,
            _ptr(self.pairs), len(self.pairs), C.c_void_p(self.knn_idx.data_ptr()), C.c_void_p(self.knn_dist.data_ptr()),
            _ptr(self.offset)))
        return self.knn_idx, self.knn_dist

    def knn2_screened(self, ratio: float):
        """Hamming only (esfm_knn2_pairs_screened_dev): the raw table of a pass that screens with `ratio` -- train index -2 marks
        the queries it dropped as unable to pass d0 < ratio d1."""
        torch = self.torch
        b = self.bank
        idx = torch.empty((max(self.total_queries, 1), 2), dtype=torch.int32, device=b.device)
        dist = torch.empty((max(self.total_queries, 1), 2), dtype=torch.float32, device=b.device)
        check(lib().esfm_knn2_pairs_screened_dev(
            self.ctx.handle, b.metric, C.c_void_p(b.data.data_ptr()), _ptr(b.row_offset), b.n_sets, b.width,
            _ptr(self.pairs), len(self.pairs), float(ratio), C.c_void_p(idx.data_ptr()), C.c_void_p(dist.data_ptr()), _ptr(self.offset)))
        return idx, dist

    def set_l2_audit(self, mode: int) -> None:
        """Certificate audit (tests): 0 product path, 1 skip the exact re-scan, 2 brute-force every query, 3 the one-product pass alone (esfm.h)."""
        check(lib().esfm_ctx_set_l2_audit(self.ctx.handle, int(mode)))

    def flagged(self) -> np.ndarray:
        """(pair index, query row) of the queries the last L2 call could not certify; synchronises."""
        n = C.c_int64(0)
        check(lib().esfm_match_last_flagged(self.ctx.handle, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 2), np.int32)
        check(lib().esfm_match_last_flagged(self.ctx.handle, _ptr(out), n.value, C.byref(n)))
        return out[:n.value].copy()

    def stats(self) -> Tuple[int, int]:
        """(queries, queries re-scanned exactly) of the last L2 call; synchronises."""
        a = C.c_int64(0); b = C.c_int64(0)
        check(lib().esfm_match_last_stats(self.ctx.handle, C.byref(a), C.byref(b)))
        return a.value, b.value


    def second_pass(self) -> int:
        """Queries of the last L2 call the one-product bf16 pass could not certify and handed to the threshold-filter stage of l2_finish_kernel (64-float descriptors); synchronises."""
        a = C.c_int64(0)
        check(lib().esfm_match_last_second_pass(self.ctx.handle, C.byref(a)))
        return a.value

def shard_pair_list(n_frames: int, rows_per_frame: Optional[np.ndarray], rank: int, world: int) -> np.ndarray:
    """This rank's share of the (i, j<i) pair list (host-only, no GPU)."""
    cap = n_frames * (n_frames - 1) // 2
    out = np.zeros((max(cap, 1), 2), np.int32)
    rows = None if rows_per_frame is None else np.ascontiguousarray(rows_per_frame, np.int32)
    n = lib().esfm_shard_pair_list(int(n_frames), _ptr(rows) if rows is not None else None, int(rank), int(world), _ptr(out))
    if n < 0:
        check(n)
    return out[:n].copy()

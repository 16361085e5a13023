"""The body of the reference's ``main`` after feature extraction (cpp_code/test/sfm.cpp:128-339), mirrored over the GPU
stages of this package: all-pairs matching -> 5-point RANSAC + relative depth per pair -> track ids -> initial pair ->
triangulation -> BA -> (next frame by PnP -> triangulation -> periodic BA)* -> final BA -> SOR filter -> .ply.

Feature extraction (detectFeaturesSURF / detectFeaturesORB from pixels, sfm.cpp:84-126) lives in
``features.py``; ``run_sfm`` starts from frames that already carry keypoints and descriptors.  The stages that the reference runs pair by
pair but whose results do not depend on the loop state (matching, RANSAC, pose recovery, depth) are batched over all pairs;
the track bookkeeping that does depend on it runs in the reference's order."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from ._lib import Context, ESFM_HAMMING, ESFM_L2_F32, default_context
from .ba import BundleAdjustment
from .cloud import CProceesing, write_ply
from .matching import DescriptorBank, FeatureMatching, PairMatcher
from .motion import MotionEstimator, _dehomogenise, find_essential_pairs, pixel2cam, recover_pose_pairs, triangulate_pairs
from .types import DMatch, Frame, SparsePointCloud


@dataclass
class FramePair:
    """frame_pair_t (utility.h:57-78)."""
    frame_id_1: int
    frame_id_2: int
    matches: List[DMatch] = field(default_factory=list)
    T_21: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))
    appro_depth: float = 1.0


def match_and_verify_all_pairs(frames: Sequence[Frame], use_feature: str = "S", ransac_reproj_distance: float = 1.0,
                               num_min_pair: int = 20, ctx: Optional[Context] = None) -> List[List[FramePair]]:
    """sfm.cpp:140-167 for every (i, j < i): matchFeatures{SURF,ORB}(frames[i], frames[j]); if more than num_min_pair
    matches survive, estimate2D2D_E5P_RANSAC (threshold = ransac_reproj_distance, prob 0.99) and getDepthFast on the inliers;
    otherwise no inliers, identity transform, depth 1 (:147-148).  Returns img_match_graph[i][j]."""
    ctx = ctx or default_context()
    n = len(frames)
    metric = ESFM_HAMMING if use_feature == "O" else ESFM_L2_F32
    ratio = 0.8 if use_feature == "O" else 0.5
    pairs = np.array([(i, j) for i in range(n) for j in range(i)], np.int32).reshape(-1, 2)
    graph: List[List[FramePair]] = [[FramePair(i, j) for j in range(i)] for i in range(n)]
    if len(pairs) == 0:
        return graph
    bank = DescriptorBank([f.descriptors for f in frames], metric, device=f"cuda:{ctx.device}")
    res = PairMatcher(bank, pairs, ctx).match(ratio).to_host()      # PairMatcher drains torch's upload stream before its first launch
    # RANSAC + pose for every pair with enough matches, in shared launches
    sel = [k for k in range(len(pairs)) if len(res[k][0]) > num_min_pair]
    if sel:
        off = np.concatenate([[0], np.cumsum([len(res[k][0]) for k in sel])]).astype(np.int32)
        p1 = np.concatenate([np.asarray(frames[pairs[k][0]].keypoints, np.float32).reshape(-1, 2)[res[k][0]] for k in sel])
        p2 = np.concatenate([np.asarray(frames[pairs[k][1]].keypoints, np.float32).reshape(-1, 2)[res[k][1]] for k in sel])
        K4 = np.array([[frames[pairs[k][0]].K_cam[0, 0], frames[pairs[k][0]].K_cam[0, 2], frames[pairs[k][0]].K_cam[1, 1],
                        frames[pairs[k][0]].K_cam[1, 2]] for k in sel], np.float32)
        Es, mask, status, _ = find_essential_pairs(off, p1, p2, K4, 0.99, ransac_reproj_distance, ctx)
        good, Rs, ts, _ = recover_pose_pairs(off, p1, p2, K4, Es, mask, ctx)
        # getDepthFast: every 20th inlier match, identity vs T_21, frame i's K for both images
        jobs_P2, jobs_a, jobs_b, jobs_off, jobs_k = [], [], [], [0], []
        for s, k in enumerate(sel):
            i, j = pairs[k]
            g = graph[i][j]
            if not status[s]:
                continue
            m = mask[off[s]:off[s + 1]]
            q, t, d = res[k]
            g.matches = [DMatch(int(a), int(b), float(c)) for a, b, c in zip(q[m], t[m], d[m])]
            T = np.eye(4, dtype=np.float32)
            T[:3, :3] = Rs[s].astype(np.float32); T[:3, 3] = ts[s].astype(np.float32)
            g.T_21 = T
            pick = np.arange(0, int(m.sum()), 20)
            if len(pick):
                Ki = frames[i].K_cam
                jobs_P2.append(T[:3]); jobs_k.append((i, j))
                jobs_a.append(pixel2cam(p1[off[s]:off[s + 1]][m][pick], Ki)); jobs_b.append(pixel2cam(p2[off[s]:off[s + 1]][m][pick], Ki))
                jobs_off.append(jobs_off[-1] + len(pick))
        if jobs_k:
            P1 = np.tile(np.eye(4, dtype=np.float32)[:3], (len(jobs_k), 1, 1))
            h = triangulate_pairs(P1, np.stack(jobs_P2), np.array(jobs_off, np.int32), np.concatenate(jobs_a), np.concatenate(jobs_b), ctx)
            for s, (i, j) in enumerate(jobs_k):
                p = _dehomogenise(h[jobs_off[s]:jobs_off[s + 1]])
                depth_sum = 0.0
                for v in p:
                    depth_sum += float(np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))
                graph[i][j].appro_depth = depth_sum / len(p)
    return graph


def propagate_track_ids(frames: Sequence[Frame], graph: List[List[FramePair]]):
    """sfm.cpp:173-216: a frame's keypoint takes the track id of its verified match in an earlier frame unless that id is
    already used in the frame; unmatched keypoints get fresh ids.  Returns (feature_track_matrix [n_frames, n_total_keypoints]
    bool, number of unique points)."""
    n = len(frames)
    total = sum(len(f.keypoints) for f in frames)
    track = np.zeros((n, max(total, 1)), bool)
    cur = 0
    for i in range(n):
        fi = frames[i]
        used = set(int(v) for v in fi.unique_pixel_ids if v >= 0)
        for j in range(i):
            fj = frames[j]
            for m in graph[i][j].matches:
                tid = int(fj.unique_pixel_ids[m.trainIdx])
                if fi.unique_pixel_ids[m.queryIdx] < 0 or fi.unique_pixel_ids[m.queryIdx] != tid:
                    if tid not in used:                                  # is_duplicated scan (:181-188)
                        old = int(fi.unique_pixel_ids[m.queryIdx])
                        fi.unique_pixel_ids[m.queryIdx] = tid
                        fi.unique_pixel_has_match[m.queryIdx] = True
                        used.add(tid)
                        if old >= 0 and not np.any(fi.unique_pixel_ids == old):
                            used.discard(old)
        fresh = 0
        for k in range(len(fi.unique_pixel_ids)):
            if fi.unique_pixel_ids[k] < 0:
                fi.unique_pixel_ids[k] = cur + fresh
                fresh += 1
            track[i, fi.unique_pixel_ids[k]] = True
        cur += fresh
    return track, cur


def run_sfm(frames: List[Frame], output_file: Optional[str] = None, use_feature: str = "S", ransac_reproj_distance: float = 1.0,
            use_track_frames_as_init: bool = True, fix_calib_tolerance_BA: float = 0.0, frequency_BA: int = 4,
            ctx: Optional[Context] = None, verbose: bool = False):
    """sfm.cpp:128-339.  Returns (sparse cloud before the final filter, filtered cloud, img_match_graph)."""
    ctx = ctx or default_context()
    fm, ee = FeatureMatching(ctx), MotionEstimator(ctx)
    for f in frames:
        f.init_pixel_ids()
    graph = match_and_verify_all_pairs(frames, use_feature, ransac_reproj_distance, 20, ctx)
    track, n_unique = propagate_track_ids(frames, graph)
    if verbose:
        for i in range(len(frames)):
            for j in range(i):
                if graph[i][j].matches:
                    print(f"Pair ( {i} , {j} ): [{len(graph[i][j].matches)}] verified matches.")
        print(f"The total unique feature point number is {n_unique}")
    init_1, init_2, depth_init = 1, 0, 10.0
    if use_track_frames_as_init:
        found, a, b, d = fm.findInitializeFramePair(track, frames, [[p.appro_depth for p in row] + [0.0] * (len(frames) - len(row)) for row in graph])
        init_1, init_2 = a, b
        if found:
            depth_init = d
    if verbose:
        print(f"Initialization frames: [ {init_1} ] and [ {init_2} ]")
    cloud = SparsePointCloud()
    frames[init_1].pose_cam = np.eye(4, dtype=np.float32)
    frames[init_2].pose_cam = (graph[init_1][init_2].T_21 @ frames[init_1].pose_cam).astype(np.float32)
    ee.doTriangulation(frames[init_1], frames[init_2], graph[init_1][init_2].matches, cloud, rgb_image=frames[init_1].rgb_image)
    todo = [True] * len(frames)
    todo[init_1] = todo[init_2] = False
    ba = BundleAdjustment(ctx)
    ba.doSFMBA(frames, todo, cloud, fix_calib_tolerance_BA)
    remaining = len(frames) - 2
    reproj = ransac_reproj_distance
    while remaining > 0:
        nxt = fm.findNextFrame(track, todo, cloud.unique_point_ids, -1)
        if nxt < 0:
            break                                                        # the reference would index with an uninitialised value
        ok = ee.estimate2D3D_P3P_RANSAC(frames[nxt], cloud, reproj)
        reproj += 1.0
        for i in range(len(frames)):
            if not todo[i]:
                if nxt > i:
                    ee.doTriangulation(frames[nxt], frames[i], graph[nxt][i].matches, cloud, rgb_image=frames[nxt].rgb_image)
                else:
                    ee.doTriangulation(frames[i], frames[nxt], graph[i][nxt].matches, cloud, rgb_image=frames[i].rgb_image)
        if not ok:
            ee.outlierFilter(cloud)
        todo[nxt] = False
        remaining -= 1
        if remaining % frequency_BA == 0:
            ba.initBA()
            ba.doSFMBA(frames, todo, cloud, fix_calib_tolerance_BA)
            reproj = ransac_reproj_distance
        if verbose:
            print(f"Progress: [ {len(frames) - remaining} / {len(frames)} ]")
    ba.doSFMBA(frames, todo, cloud)
    out = CProceesing(ctx).SORFilter(cloud)
    if output_file:
        write_ply(output_file, out)
    return cloud, out, graph

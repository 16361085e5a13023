"""Seeded synthetic workloads for the hot path.

Measurement/test infrastructure: the reference ships no benchmark inputs, so
bench.py and the tests build descriptor sets and BA scenes from these recipes.
All generators are numpy ``default_rng(PCG64(seed))``; nothing here touches the GPU.

  M-SURF-4k  25 x 4096 x 64 f32   surf_like_sets(25, 4096, pool=16384, seed_base=1000)
  M-SURF-8k  256 x 8192 x 64 f32  surf_like_sets(256, 8192, pool=65536, seed_base=2000)
  M-ORB-4k   25 x 4096 x 32 u8    orb_like_sets(25, 4096, pool=16384, seed_base=3000)
  M-SURF-4k-hard 25 x 4096 x 64   msurf4k_hard_sets(SURF-300 descriptors of the fountain images): real, clustered descriptors resampled
  BA-25      25 cams, 30k pts, 8 obs/pt    ba_scene(25, 30000, 8, radius=10, extent=2, seed=4000)
  BA-512     512 cams, 300k pts, 10 obs/pt ba_scene(512, 300000, 10, radius=40, extent=8, seed=5000)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

# fx, cx, fy, cy of test_data/k_25/K.txt (reference fountain set, 768x512)
FOUNTAIN_K4 = (689.87, 380.17, 691.04, 251.70)


def _unit_rows(x: np.ndarray) -> np.ndarray:
    n = np.sqrt((x.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
    return (x / np.maximum(n, 1e-30)).astype(np.float32)


def surf_like_sets(n_images: int, n_feats: int, pool: int = 16384, seed_base: int = 1000,
                   dim: int = 64, noise: float = 0.05) -> List[np.ndarray]:
    """SURF-like float descriptors: rows L2-normalised (SURF's are,
    feature_matching.cpp:45-52 [upstream]); half of each image re-observes a
    shared track pool with N(0, noise^2) perturbation, half is fresh."""
    pool_vecs = _unit_rows(np.random.default_rng(np.random.PCG64(seed_base - 1)).standard_normal((pool, dim)))
    sets = []
    for i in range(n_images):
        rng = np.random.default_rng(np.random.PCG64(seed_base + i))
        n_trk = n_feats // 2
        ids = rng.choice(pool, size=min(n_trk, pool), replace=False)
        a = pool_vecs[ids].astype(np.float64) + noise * rng.standard_normal((len(ids), dim))
        b = rng.standard_normal((n_feats - len(ids), dim))
        d = _unit_rows(np.concatenate([a, b], axis=0))
        rng.shuffle(d, axis=0)
        sets.append(np.ascontiguousarray(d, np.float32))
    return sets


def surf_resampled_sets(desc_pool: np.ndarray, n_images: int, n_feats: int, seed_base: int = 6000, track_frac: float = 0.5,
                        track_noise: float = 0.02, fresh_noise: float = 0.06) -> List[np.ndarray]:
    """"M-SURF-4k-hard": REAL SURF descriptors resampled to the metric's shape.  `desc_pool` holds the descriptors a SURF detector
    produced on real images (bench.py / the tests: the reference's 11 fountain images at minHessian 300, ~30 k rows) -- clustered,
    with repeated structure (window frames, bricks), which is what makes the second nearest neighbour close to the first and a
    bf16 candidate list hard to certify; surf_like_sets' "fresh" rows are isotropic and give the ratio screen nothing to decide.
    Every image takes `track_frac` of its rows from a shared set of `n_feats` pool rows (the tracks: the same scene point seen
    again, perturbed by N(0, track_noise^2) per component and re-normalised) and the rest from other pool rows perturbed by
    `fresh_noise` (near real descriptors, not copies).  Rows shuffled.  Deterministic in (pool, seed_base)."""
    pool = np.ascontiguousarray(desc_pool, np.float32)
    n_pool, dim = pool.shape
    base = np.random.default_rng(np.random.PCG64(seed_base - 1))
    track_ids = base.choice(n_pool, size=min(n_feats, n_pool), replace=False)
    sets = []
    for i in range(n_images):
        rng = np.random.default_rng(np.random.PCG64(seed_base + i))
        n_trk = int(n_feats * track_frac)
        ids = rng.choice(track_ids, size=min(n_trk, len(track_ids)), replace=False)
        a = pool[ids].astype(np.float64) + track_noise * rng.standard_normal((len(ids), dim))
        other = rng.choice(n_pool, size=n_feats - len(ids), replace=n_pool < n_feats)
        b = pool[other].astype(np.float64) + fresh_noise * rng.standard_normal((len(other), dim))
        d = _unit_rows(np.concatenate([a, b], axis=0))
        rng.shuffle(d, axis=0)
        sets.append(np.ascontiguousarray(d, np.float32))
    return sets


HARD_POOL_ROWS, HARD_TRACK_NOISE, HARD_FRESH_NOISE = 8192, 0.005, 0.01


def msurf4k_hard_sets(fountain_descriptors: np.ndarray, n_images: int = 25, n_feats: int = 4096) -> List[np.ndarray]:
    """Workload "M-SURF-4k-hard" (bench.py's `hard` leg, tests/test_metric_workloads_gpu.py): surf_resampled_sets on the first 8192
    SURF descriptors of the reference's fountain images (minHessian 300, image order), seeds 6000 + image.  Measured on MI355X at the
    reference's ratio 0.5: 34 % of the queries survive the matcher's ratio screen (M-SURF-4k: 3.7 %), 1.4 % reach its second pass
    (M-SURF-4k: none); at ratio 0.8: 62 % and 3.3 %."""
    return surf_resampled_sets(np.ascontiguousarray(fountain_descriptors[:HARD_POOL_ROWS]), n_images, n_feats, seed_base=6000,
                               track_noise=HARD_TRACK_NOISE, fresh_noise=HARD_FRESH_NOISE)


def orb_like_sets(n_images: int, n_feats: int, pool: int = 16384, seed_base: int = 3000,
                  nbytes: int = 32, flip: float = 0.08) -> List[np.ndarray]:
    """ORB-like 256-bit descriptors: half re-observes a pool with each bit flipped w.p. `flip`."""
    pool_bits = np.random.default_rng(np.random.PCG64(seed_base - 1)).integers(0, 2, (pool, nbytes * 8), dtype=np.uint8)
    sets = []
    for i in range(n_images):
        rng = np.random.default_rng(np.random.PCG64(seed_base + i))
        n_trk = n_feats // 2
        ids = rng.choice(pool, size=min(n_trk, pool), replace=False)
        a = pool_bits[ids] ^ (rng.random((len(ids), nbytes * 8)) < flip).astype(np.uint8)
        b = rng.integers(0, 2, (n_feats - len(ids), nbytes * 8), dtype=np.uint8)
        bits = np.concatenate([a, b], axis=0)
        rng.shuffle(bits, axis=0)
        sets.append(np.ascontiguousarray(np.packbits(bits, axis=1), np.uint8))
    return sets


def all_pairs(n_frames: int) -> np.ndarray:
    """The (i, j<i) list of the pair loop cpp_code/test/sfm.cpp:140-143: query = i, train = j."""
    return np.array([(i, j) for i in range(n_frames) for j in range(i)], np.int32).reshape(-1, 2)


# ----------------------------------------------------------------------------- BA
def _rodrigues_to_aa(R: np.ndarray) -> np.ndarray:
    """Rotation matrix -> angle-axis (what cv::Rodrigues does at ba.cpp:82)."""
    c = (np.trace(R) - 1.0) / 2.0
    theta = np.arccos(np.clip(c, -1.0, 1.0))
    if theta < 1e-12:
        return np.zeros(3)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2.0 * np.sin(theta))
    return w * theta


def aa_to_R(aa: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(aa)
    if th < 1e-15:
        return np.eye(3)
    k = aa / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


@dataclass
class BAScene:
    cam_idx: np.ndarray   # int32 [n_obs]
    pt_idx: np.ndarray    # int32 [n_obs]
    uv: np.ndarray        # float32 [n_obs, 2]  (points_2d_, ba.cpp:37)
    K4: np.ndarray        # float32 [n_cam, 4]  fx, cx, fy, cy
    cams0: np.ndarray     # float64 [n_cam, 6]  perturbed start (angle-axis, t)
    pts0: np.ndarray      # float64 [n_pt, 3]
    cams_gt: np.ndarray
    pts_gt: np.ndarray

    @property
    def n_cam(self): return self.cams0.shape[0]
    @property
    def n_pt(self): return self.pts0.shape[0]
    @property
    def n_obs(self): return self.cam_idx.shape[0]


def ba_scene(n_cam: int, n_pt: int, obs_per_pt: int, radius: float = 10.0, extent: float = 2.0,
             seed: int = 4000, uv_noise: float = 0.5, outlier_frac: float = 0.02,
             start_noise: Tuple[float, float, float] = (0.01, 0.05, 0.05),
             camera_major: bool = True) -> BAScene:
    """Ring of cameras looking at the origin; each point seen by `obs_per_pt`
    consecutive ring cameras.  Observation order is camera-major, point-minor,
    as setBAProblem emits it (ba.cpp:22-48) unless camera_major=False."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    obs_per_pt = min(obs_per_pt, n_cam)
    cams = np.zeros((n_cam, 6))
    for c in range(n_cam):
        ang = 2 * np.pi * c / n_cam
        C = np.array([radius * np.cos(ang), 0.3 * radius * np.sin(2 * ang) * 0.2, radius * np.sin(ang)])
        z = -C / np.linalg.norm(C)                      # optical axis towards origin
        x = np.cross(np.array([0.0, 1.0, 0.0]), z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z], axis=0)                 # world -> camera
        cams[c, :3] = _rodrigues_to_aa(R)
        cams[c, 3:] = -R @ C
    pts = rng.uniform(-extent, extent, size=(n_pt, 3))
    start = rng.integers(0, n_cam, size=n_pt)
    cam_idx = ((start[:, None] + np.arange(obs_per_pt)[None, :]) % n_cam).astype(np.int32)
    pt_idx = np.repeat(np.arange(n_pt, dtype=np.int32)[:, None], obs_per_pt, axis=1)
    cam_idx = cam_idx.reshape(-1); pt_idx = pt_idx.reshape(-1)
    if camera_major:
        order = np.lexsort((pt_idx, cam_idx))
        cam_idx, pt_idx = cam_idx[order], pt_idx[order]
    K4 = np.tile(np.array(FOUNTAIN_K4, np.float32), (n_cam, 1))
    # project
    Rs = np.stack([aa_to_R(cams[c, :3]) for c in range(n_cam)])
    P = np.einsum("nij,nj->ni", Rs[cam_idx], pts[pt_idx]) + cams[cam_idx, 3:]
    uv = np.stack([P[:, 0] / P[:, 2] * K4[0, 0] + K4[0, 1], P[:, 1] / P[:, 2] * K4[0, 2] + K4[0, 3]], axis=1)
    uv += uv_noise * rng.standard_normal(uv.shape)
    n_out = int(outlier_frac * len(uv))
    if n_out > 0:
        sel = rng.choice(len(uv), size=n_out, replace=False)
        uv[sel] += rng.uniform(-50, 50, size=(n_out, 2))
    cams0 = cams.copy()
    cams0[:, :3] += start_noise[0] * rng.standard_normal((n_cam, 3))
    cams0[:, 3:] += start_noise[1] * rng.standard_normal((n_cam, 3))
    pts0 = pts + start_noise[2] * rng.standard_normal(pts.shape)
    return BAScene(cam_idx=np.ascontiguousarray(cam_idx, np.int32), pt_idx=np.ascontiguousarray(pt_idx, np.int32),
                   uv=np.ascontiguousarray(uv, np.float32), K4=np.ascontiguousarray(K4, np.float32),
                   cams0=cams0, pts0=pts0, cams_gt=cams, pts_gt=pts)


def in_reference_frame(sc: BAScene, ref: int) -> BAScene:
    """Re-express the scene in camera `ref`'s frame, so that camera `ref` sits at the origin with zero rotation --
    how the reference's pipeline holds its reference frame (ba.cpp:155-162 bounds its pose to +-1e-10).
    The start values are moved with the START pose of `ref`, the ground truth with its true pose."""
    def move(cams, pts):
        R0, t0 = aa_to_R(cams[ref, :3]), cams[ref, 3:]
        out_c = np.zeros_like(cams)
        for c in range(cams.shape[0]):
            Rc = aa_to_R(cams[c, :3]) @ R0.T
            out_c[c, :3] = _rodrigues_to_aa(Rc)
            out_c[c, 3:] = cams[c, 3:] - Rc @ t0
        out_c[ref] = 0.0
        return out_c, pts @ R0.T + t0
    cams0, pts0 = move(sc.cams0, sc.pts0)
    cams_gt, pts_gt = move(sc.cams_gt, sc.pts_gt)
    return BAScene(cam_idx=sc.cam_idx, pt_idx=sc.pt_idx, uv=sc.uv, K4=sc.K4, cams0=cams0, pts0=pts0,
                   cams_gt=cams_gt, pts_gt=pts_gt)


def sfm_scene(n_cam: int = 8, n_pt: int = 900, seed: int = 7000, pix_noise: float = 0.3, n_clutter: int = 150, desc_noise: float = 0.03):
    """A synthetic reconstruction problem for the whole pipeline (cpp_code/test/sfm.cpp:128-339): cameras on an arc looking at
    a point cloud with depth relief; every camera sees the points in front of it inside the 768 x 512 image; a keypoint's
    descriptor is its point's random unit 64-vector plus noise (SURF-like, L2-normalised); clutter keypoints carry fresh
    random descriptors.  Returns (frames as dicts of numpy arrays: keypoints, descriptors, point_id per keypoint; K [3,3];
    camera poses world->camera [n_cam,4,4]; points [n_pt,3])."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    K = np.array([[FOUNTAIN_K4[0], 0, FOUNTAIN_K4[1]], [0, FOUNTAIN_K4[2], FOUNTAIN_K4[3]], [0, 0, 1]], np.float32)
    pts = np.stack([rng.uniform(-3, 3, n_pt), rng.uniform(-2, 2, n_pt), rng.uniform(-1.5, 1.5, n_pt)], 1)
    base = rng.standard_normal((n_pt, 64)); base /= np.linalg.norm(base, axis=1, keepdims=True)
    poses, frames = [], []
    for c in range(n_cam):
        ang = (c - (n_cam - 1) / 2) * 0.16
        C = np.array([9.0 * np.sin(ang), 0.15 * np.cos(3 * ang), -9.0 * np.cos(ang)])
        z = -C / np.linalg.norm(C)
        x = np.cross(np.array([0.0, 1.0, 0.0]), z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z], 0)
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = -R @ C
        poses.append(T)
        Xc = pts @ R.T + T[:3, 3]
        uv = np.stack([Xc[:, 0] / Xc[:, 2] * K[0, 0] + K[0, 2], Xc[:, 1] / Xc[:, 2] * K[1, 1] + K[1, 2]], 1)
        vis = (Xc[:, 2] > 1) & (uv[:, 0] > 5) & (uv[:, 0] < 763) & (uv[:, 1] > 5) & (uv[:, 1] < 507)
        ids = np.nonzero(vis)[0]
        ids = ids[rng.random(len(ids)) < 0.9]                        # detector misses
        kp = uv[ids] + pix_noise * rng.standard_normal((len(ids), 2))
        d = base[ids] + desc_noise * rng.standard_normal((len(ids), 64))
        ck = np.stack([rng.uniform(5, 763, n_clutter), rng.uniform(5, 507, n_clutter)], 1)
        cd = rng.standard_normal((n_clutter, 64))
        kp = np.concatenate([kp, ck]); d = np.concatenate([d, cd]); pid = np.concatenate([ids, -np.ones(n_clutter, np.int64)])
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        order = rng.permutation(len(kp))
        frames.append(dict(keypoints=kp[order].astype(np.float32), descriptors=np.ascontiguousarray(d[order], np.float32), point_id=pid[order]))
    return frames, K, np.stack(poses), pts

"""Multi-rank behaviour of the two shardable stages, world sizes 2 and 4 over gloo on CPU.

The HIP kernels cannot run here, so the per-rank compute is stood in for by the oracle (test
infrastructure); what is under test is the decomposition the product uses:
  * matching: the rank shards of the pair list are disjoint, complete, and concatenating per-shard
    results reproduces the single-rank result -- no collective on the data path;
  * BA: points (with their observations) are partitioned; per-rank partial reduced camera systems
    SUM-all-reduce to the unsharded system (the one exchange step of SURVEY 8e), as do the per-camera
    column norms that define the LM diagonal.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import easysfm_amd as E
        import oracle
        from easysfm_amd import synth
        oracle.set_num_threads(1)
        # ---- matching: pair-list partition, no collective on the data path
        sets = synth.surf_like_sets(5, 120, pool=128, seed_base=50)
        rows = np.array([len(s) for s in sets], np.int32)
        mine = E.shard_pair_list(5, rows, rank, world)
        local = {(int(i), int(j)): oracle.match_l2(sets[i], sets[j], 0.8) for i, j in mine}
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: tuple(a.tolist() for a in v) for k, v in local.items()})
        if rank == 0:
            merged = {}
            for g in gathered:
                assert not (set(g) & set(merged)); merged.update(g)
            assert sorted(merged) == sorted((i, j) for i in range(5) for j in range(i))
            for (i, j), v in merged.items():
                ref = oracle.match_l2(sets[i], sets[j], 0.8)
                assert all(np.array_equal(np.array(a), b) for a, b in zip(v, ref))
        # ---- BA: partial reduced systems sum to the full one
        sc = synth.ba_scene(6, 90, 4, seed=60)
        shard = E.shard_points(sc.n_pt, sc.pt_idx, world)
        keep = shard[sc.pt_idx] == rank
        nc, npp = oracle.ba_column_sqnorms(sc.n_cam, sc.n_pt, sc.cam_idx[keep], sc.pt_idx[keep], sc.uv[keep], sc.K4, sc.cams0, sc.pts0, 0.5)
        t = torch.from_numpy(nc); dist.all_reduce(t)          # camera column norms: SUM over ranks
        t2 = torch.from_numpy(npp); dist.all_reduce(t2)       # (points are owned by one rank: sum = union)
        nc_full, np_full = oracle.ba_column_sqnorms(sc.n_cam, sc.n_pt, sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, 0.5)
        assert np.allclose(nc, nc_full, rtol=1e-12) and np.allclose(npp, np_full, rtol=1e-12)
        diag_c = np.clip(nc, 1e-6, 1e32); diag_p = np.clip(npp, 1e-6, 1e32)
        S, rhs = oracle.ba_partial_reduced(sc.n_cam, sc.n_pt, sc.cam_idx[keep], sc.pt_idx[keep], sc.uv[keep], sc.K4, sc.cams0, sc.pts0,
                                           0.5, 1e4, diag_c, diag_p, add_cam_diag=(rank == 0))
        buf = torch.from_numpy(np.concatenate([S.ravel(), rhs]))
        dist.all_reduce(buf)                                    # the one exchange step per LM iteration
        Sf, rf = oracle.ba_partial_reduced(sc.n_cam, sc.n_pt, sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0,
                                           0.5, 1e4, diag_c, diag_p, add_cam_diag=True)
        got = buf.numpy()
        assert np.allclose(got[:S.size].reshape(S.shape), Sf, rtol=1e-10, atol=1e-9 * np.abs(Sf).max())
        assert np.allclose(got[S.size:], rf, rtol=1e-10, atol=1e-9 * np.abs(rf).max())
        y = np.linalg.solve(got[:S.size].reshape(S.shape), got[S.size:])          # every rank solves the same system
        ys = [None] * world
        dist.all_gather_object(ys, y.tolist())
        assert all(np.array_equal(np.array(v), np.array(ys[0])) for v in ys)       # bit-identical camera steps on all ranks
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharding_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res

"""Observation-sharded bundle adjustment with the real HIP kernels on every rank.

The GPU box has one MI355X, so the 2-rank case runs both ranks on cuda:0 over the gloo backend
(RCCL refuses two ranks on one device); the collective is issued by the product's own
esfm_allreduce_fn implementation (easysfm_amd.ba.torch_allreduce_callback) on device buffers.
A 1-rank NCCL (= RCCL) group exercises the backend bench.py uses at N > 1.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


CALIB0 = [703.67, 376.37, 677.22, 254.22]     # shared intrinsics start, ~2 % off the scene's


SCENES = {"small": (8, 600, 5, 21, 10.0, 2.0), "loop200": (200, 12000, 8, 34, 15.0, 3.0)}     # n_cam, n_pt, obs per point, seed, radius, extent


def _scene(name):
    from easysfm_amd import synth
    n_cam, n_pt, k, seed, radius, extent = SCENES[name]
    return synth.ba_scene(n_cam, n_pt, k, radius=radius, extent=extent, seed=seed)


def _worker(rank, world, port, backend, q, constrained=False, scene="small"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    try:
        torch.cuda.set_device(0)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        import easysfm_amd as E
        from easysfm_amd import synth
        sc = _scene(scene)
        shard = E.shard_points(sc.n_pt, sc.pt_idx, world)
        keep = shard[sc.pt_idx] == rank
        ctx = E.Context.on_torch_stream(0)
        opt = E.default_options(); opt.max_num_iterations = 6
        cal = None
        with torch.cuda.stream(ctx.torch_stream):
            if constrained:
                sc = synth.in_reference_frame(sc, 0)
                opt.max_num_iterations = 8
                cams, pts, cal, summ = E.ba_solve_ex(sc.cam_idx[keep], sc.pt_idx[keep], sc.uv[keep], None, sc.cams0, sc.pts0,
                                                     calib=CALIB0, calib_tol=8.0, ref_cam=0, options=opt, ctx=ctx,
                                                     allreduce=E.torch_allreduce_callback())
            else:
                cams, pts, summ = E.ba_solve(sc.cam_idx[keep], sc.pt_idx[keep], sc.uv[keep], sc.K4, sc.cams0, sc.pts0, opt, ctx,
                                             allreduce=E.torch_allreduce_callback())
        q.put((rank, "ok", cams, pts, [it.cost for it in summ.log()], summ.num_active_points, cal,
               [it.line_search_steps for it in summ.log()]))
        dist.barrier()
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), None, None, None, None, None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _run(world, backend, constrained=False, scene="small"):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q, constrained, scene)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    return sorted(res, key=lambda r: r[0])


def _single(scene="small"):
    import easysfm_amd as E
    sc = _scene(scene)
    opt = E.default_options(); opt.max_num_iterations = 6
    return sc, E.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, opt, E.Context(0, None))


def test_sharded_ba_two_ranks_matches_single(gpu_ctx):
    sc, (cams, pts, summ) = _single()
    res = _run(2, "gloo")
    ref_cost = [it.cost for it in summ.log()]
    for rank, _, c, p, costs, n_active, _cal, _ls in res:
        assert len(costs) == len(ref_cost)
        assert np.allclose(costs, ref_cost, rtol=1e-9)               # every rank sees the global cost trace
        assert np.allclose(c, cams, rtol=1e-6, atol=1e-6)             # replicated cameras
        assert np.allclose(p, pts, rtol=1e-6, atol=1e-6)              # and, after the final merge, ALL points
        assert 0 < n_active < sc.n_pt                                  # but owns only its shard
    assert np.array_equal(res[0][2], res[1][2])                       # camera blocks bit-identical across ranks
    assert res[0][5] + res[1][5] == sc.n_pt


def test_rccl_backend_single_rank(gpu_ctx):
    """The nccl (RCCL) process group + the device-pointer callback, world size 1."""
    sc, (cams, pts, summ) = _single()
    (rank, _, c, p, costs, n_active, _cal, _ls), = _run(1, "nccl")
    assert np.allclose(costs, [it.cost for it in summ.log()], rtol=1e-12)
    assert np.allclose(c, cams, rtol=1e-9, atol=1e-12) and np.allclose(p, pts, rtol=1e-9, atol=1e-12)


def test_sharded_constrained_ba_two_ranks_matches_single(gpu_ctx):
    """Free shared intrinsics (tight box, so the line search contracts) + reference camera, observations sharded over two
    ranks: the intrinsics block rides in the all-reduced reduced system and F'F sums, the line-search scalars (slope, max
    |delta|) in the SUM / MAX scalar reductions -- every rank must take the same decisions as the single-GPU solve."""
    import easysfm_amd as E
    from easysfm_amd import synth
    sc = synth.in_reference_frame(synth.ba_scene(8, 600, 5, seed=21), 0)
    opt = E.default_options(); opt.max_num_iterations = 8
    cams, pts, cal, summ = E.ba_solve_ex(sc.cam_idx, sc.pt_idx, sc.uv, None, sc.cams0, sc.pts0, calib=CALIB0, calib_tol=8.0,
                                         ref_cam=0, options=opt, ctx=E.Context(0, None))
    res = _run(2, "gloo", constrained=True)
    ref_cost = [it.cost for it in summ.log()]
    ref_ls = [it.line_search_steps for it in summ.log()]
    for rank, _, c, p, costs, n_active, k, ls in res:
        assert ls == ref_ls
        assert np.allclose(costs, ref_cost, rtol=1e-9)
        assert np.allclose(k, cal, rtol=1e-9, atol=1e-7)
        assert np.allclose(c, cams, rtol=1e-6, atol=1e-6) and np.allclose(p, pts, rtol=1e-6, atol=1e-6)
        assert np.all(np.abs(c[0]) <= 1e-10)
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][6], res[1][6])


def test_native_rccl_communicator_single_rank(gpu_ctx):
    """The library's own exchange (esfm_comm_*: librccl bound at run time, ncclAllReduce on the context's stream, no callback
    into Python): a one-rank communicator -- RCCL refuses two ranks on one device, and the box has one -- must leave the
    solve bit-identical to the single-GPU one, through the same packed reduced-system exchange the 8-GPU run uses; and
    esfm_comm_allreduce on a device buffer is the identity."""
    import torch
    import easysfm_amd as E
    from easysfm_amd import synth
    sc, (cams, pts, summ) = _single()
    ctx = E.Context.on_torch_stream(0)
    comm = E.Comm(ctx, E.Comm.unique_id(), 0, 1)
    with torch.cuda.stream(ctx.torch_stream):
        x = torch.arange(1000, dtype=torch.float64, device="cuda:0")
        comm.allreduce_(x.data_ptr(), x.numel())
        ctx.synchronize()
        assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float64))
        opt = E.default_options(); opt.max_num_iterations = 6
        c, p, s2 = E.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, opt, ctx, allreduce=comm)
    comm.close()
    # a one-rank sum is the identity and the solver is bit-reproducible: equal, not close
    assert [it.cost for it in s2.log()] == [it.cost for it in summ.log()]
    # (cameras are replicated; the sharded solve returns points as x0 + all-reduce(x - x0), one rounding away from x)
    assert np.array_equal(c, cams) and np.allclose(p, pts, rtol=0.0, atol=4e-15)


def test_sharded_structure_aware_solve_two_ranks_matches_single(gpu_ctx):
    """A loop of 200 cameras: the reduced camera system takes the structure-aware solve (ba_chol_sparse.hip).  Two ranks hold
    half of the points each, so NEITHER rank's own observations show the whole co-visibility: the ranks unite their camera-pair
    sets at the start of the solve, plan identically, and exchange only the co-visible camera blocks per LM iteration
    (ba_sparse_pack_kernel -> all-reduce -> chol_sparse_assemble_kernel<PACKED>).  Cost trace and parameters must follow the
    single-GPU solve; the replicated cameras must agree bit for bit across the ranks."""
    import easysfm_amd as E
    sc, (cams, pts, summ) = _single("loop200")
    plan = E.reduced_plan(sc.n_cam, sc.n_pt, sc.cam_idx, sc.pt_idx)
    assert plan["worthwhile"]                                          # (this test is about the structure-aware path)
    shard = E.shard_points(sc.n_pt, sc.pt_idx, 2)
    for r in range(2):                                                  # a rank's own pair set is smaller than the union
        keep = shard[sc.pt_idx] == r
        assert E.reduced_plan(sc.n_cam, sc.n_pt, sc.cam_idx[keep], sc.pt_idx[keep])["covisible_blocks"] <= plan["covisible_blocks"]
    res = _run(2, "gloo", scene="loop200")
    ref_cost = [it.cost for it in summ.log()]
    for rank, _, c, p, costs, n_active, _cal, _ls in res:
        assert len(costs) == len(ref_cost) and np.allclose(costs, ref_cost, rtol=1e-9)
        assert np.allclose(c, cams, rtol=1e-6, atol=1e-6) and np.allclose(p, pts, rtol=1e-6, atol=1e-6)
    assert np.array_equal(res[0][2], res[1][2])
    assert res[0][5] + res[1][5] == sc.n_pt


def test_sharded_structure_aware_solve_four_ranks_matches_single(gpu_ctx):
    """The same with FOUR ranks (four processes on the one GPU, gloo): each holds a quarter of the points, the union of the pair sets
    is counted over four ranks, the exchange sums four packed buffers.  (The driver's 8-GPU run has never taken place -- this is as
    many ranks as the one-GPU box sensibly carries.)"""
    sc, (cams, pts, summ) = _single("loop200")
    res = _run(4, "gloo", scene="loop200")
    ref_cost = [it.cost for it in summ.log()]
    for rank, _, c, p, costs, n_active, _cal, _ls in res:
        assert len(costs) == len(ref_cost) and np.allclose(costs, ref_cost, rtol=1e-9)
        assert np.allclose(c, cams, rtol=1e-6, atol=1e-6) and np.allclose(p, pts, rtol=1e-6, atol=1e-6)
    for r in range(1, 4):
        assert np.array_equal(res[0][2], res[r][2])                     # replicated cameras: bit-identical on every rank
    assert sum(r[5] for r in res) == sc.n_pt


def test_sharded_structure_aware_solve_eight_ranks_matches_single(gpu_ctx):
    """EIGHT ranks (the node's rank count; eight processes on the one GPU, gloo): an eighth of the points each, the union of the pair sets
    over eight ranks, eight packed buffers summed."""
    sc, (cams, pts, summ) = _single("loop200")
    res = _run(8, "gloo", scene="loop200")
    ref_cost = [it.cost for it in summ.log()]
    for rank, _, c, p, costs, n_active, _cal, _ls in res:
        assert len(costs) == len(ref_cost) and np.allclose(costs, ref_cost, rtol=1e-9)
        assert np.allclose(c, cams, rtol=1e-6, atol=1e-6) and np.allclose(p, pts, rtol=1e-6, atol=1e-6)
    for r in range(1, 8):
        assert np.array_equal(res[0][2], res[r][2])
    assert sum(r[5] for r in res) == sc.n_pt


def test_native_rccl_communicator_single_rank_structure_aware(gpu_ctx):
    """The structure-aware exchange through the library's own RCCL communicator, one rank: the packed co-visible blocks are this
    rank's fixed-point sums converted exactly as the one-rank assembly converts them, so the solve is bit-identical to the single-GPU one."""
    import torch
    import easysfm_amd as E
    sc, (cams, pts, summ) = _single("loop200")
    ctx = E.Context.on_torch_stream(0)
    comm = E.Comm(ctx, E.Comm.unique_id(), 0, 1)
    assert comm.rccl_ranks() == 1
    with torch.cuda.stream(ctx.torch_stream):
        opt = E.default_options(); opt.max_num_iterations = 6
        c, p, s2 = E.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, opt, ctx, allreduce=comm)
    comm.close()
    assert [it.cost for it in s2.log()] == [it.cost for it in summ.log()]
    assert np.array_equal(c, cams) and np.allclose(p, pts, rtol=0.0, atol=4e-15)

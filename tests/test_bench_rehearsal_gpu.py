"""bench.py's N-rank run REHEARSED on the one GPU of the box: `ESFM_BENCH_BACKEND=gloo` starts the
N ordinary processes the real run starts (spawned before anything touches a GPU), lets them share device 0, gives torch a gloo group and
sends the sharded BA legs' exchange through the callback over it -- RCCL refuses two ranks on one device.  Everything else is the code the
driver's 8-GPU run executes for the first time: torch's process group and the library's communicator bring-up in one process, the agreed
fall-back to torch's group, the pair-list and point shards, the config-4 / config-5 legs, rank 0's JSON assembly.  The throughput figures
of such a run (eight processes time-sharing a GPU) mean nothing and are not looked at."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(n, *flags, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ESFM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "3", "--warmup", "1", "--no-e2e", *flags], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                # rank 0 only
    return lines[0]


def test_eight_rank_rehearsal_on_one_gpu(gpu_ctx):
    import easysfm_amd as E
    from easysfm_amd import synth
    iters5 = 4
    j = _run(8, "--ba-iters", "5", "--ba512-iters", str(iters5), "--config4-steps", "1")
    assert j["n_gpus"] == 8 and j["backend"] == "gloo-rehearsal" and j["scaling"] == "strong"
    assert j["pairs_per_rank"] == [38, 38, 38, 38, 37, 37, 37, 37] and j["config"]["pairs_per_step"] == 300
    assert "torch.distributed gloo group" in j["ba_allreduce_via"] and "not attempted: 8 ranks share" in j["ba_allreduce_via"]
    # the line's front: both halves of the metric and the legs as flat scalars inside `roofline` (what the driver's record keeps)
    head = list(j["roofline"])[:24]
    for k in ("frac", "ba_lm_iters_per_s", "ba_ms_per_iteration", "ba_sweep_frac", "config4_pairs_per_s", "config5_lm_iters_per_s", "rccl_ranks"):
        assert k in head and j["roofline"][k] is not None, k
    assert list(j)[-1] == "ba"                               # ... and the BA object at the END of the line (the stdout tail)
    # config 4: every one of the 32 640 pairs matched by exactly one rank
    c4 = j["config4"]
    assert "error" not in c4 and c4["pairs_covered"] == 256 * 255 // 2 == 32640 and c4["matches_per_step"] > 0
    # BA-25 and config 5: points sharded over eight ranks, the reduced system summed through the callback -- the cost trace is the
    # single-rank solve's
    for leg, (n_cam, n_pt, k, radius, extent, seed), iters in ((j["ba"], (25, 30000, 8, 10.0, 2.0, 4000), 5), (j["config5"], (512, 300000, 10, 40.0, 8.0, 5000), iters5)):
        assert "error" not in leg and leg["n_gpus"] == 8 and leg["config"]["obs_sharded_by_point"]
        sc = synth.ba_scene(n_cam, n_pt, k, radius=radius, extent=extent, seed=seed)
        opt = E.default_options(); opt.max_num_iterations = iters
        opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
        _c, _p, summ = E.ba_solve(sc.cam_idx, sc.pt_idx, sc.uv, sc.K4, sc.cams0, sc.pts0, opt, gpu_ctx)
        ref = [it.cost for it in summ.log()]
        assert len(leg["cost_trace"]) == len(ref) and np.allclose(leg["cost_trace"], ref, rtol=1e-9, atol=0.0), (leg["cost_trace"], ref)
    assert j["config5"]["reduced_solve"]["kind"].startswith("structure-aware")


def test_forced_communicator_failure_on_one_rank_still_yields_the_line(gpu_ctx):
    """A rank whose pre-check fails (here: forced) must not leave the others inside ncclCommInitRank: the ranks agree BEFORE anyone
    calls esfm_comm_create, fall back to torch's group together, and the line says why."""
    j = _run(4, "--ba-iters", "3", "--no-config45", ESFM_BENCH_FAIL_COMM_RANK="3")
    assert j["n_gpus"] == 4 and "error" not in j["ba"] and j["ba"]["lm_iterations"] >= 3
    assert "torch.distributed" in j["ba_allreduce_via"] and "rank 3: forced failure" in j["ba_allreduce_via"]

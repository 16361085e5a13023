"""bench.py's multi-rank start-up without a GPU (VERDICT r02, next #5a): the self-launcher (`--gpus N` with no WORLD_SIZE in the
environment) and the driver's `python -m torch.distributed.run` form both reach rank 0's single JSON line with n_gpus = N.
`--dry-run` imports neither torch nor the HIP library: the children are started before anything could touch a GPU."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def _json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_self_launcher_two_ranks_dry_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"], env=_clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                      # rank 0 only
    j = lines[0]
    assert j["n_gpus"] == 2 and j["ranks_seen"] == [0, 1] and j["ok"] and j["steps"] == 3 and j["warmup"] == 1
    assert j["torch_imported"] is False
    assert j["scaling"] == "strong" and j["pairs_per_rank"] == [150, 150] and j["pairs_total"] == 300      # M-SURF-4k's own list, split
    for rk in (0, 1):
        assert f"RANK={rk} LOCAL_RANK={rk} WORLD_SIZE=2 MASTER=127.0.0.1:" in r.stderr


def test_self_launcher_propagates_child_failure():
    # a rank that cannot reach rank 0 must not leave the launcher reporting success: WORLD_SIZE mismatch -> rank 0 waits for
    # a rank that never comes, its accept() times out -> non-zero.  Cheaper to provoke: an unknown flag makes every child exit 2.
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--no-such-flag"], env=_clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode != 0


def test_torch_distributed_run_two_ranks_dry_run():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ok"], r.stdout


def test_single_rank_dry_run_is_the_metric_workload():
    r = subprocess.run([sys.executable, BENCH, "--dry-run"], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 0
    j = _json_lines(r.stdout)[0]
    assert j["n_gpus"] == 1 and j["pairs_total"] == 300


def test_self_launcher_eight_ranks_dry_run_shares():
    """What `bench.py --gpus 8` does rank by rank before anything is timed, without a GPU: eight
    ranks meet, the headline's 300 pairs fall 38 / 37 per rank, config 4's 32 640 pairs 4 080 per rank, the weak side
    leg takes 70 images (2 415 pairs >= 300 x 8)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"], env=_clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    j = _json_lines(r.stdout)[0]
    assert j["n_gpus"] == 8 and j["ranks_seen"] == list(range(8)) and j["ok"] and j["torch_imported"] is False
    assert j["pairs_per_rank"] == [38, 38, 38, 38, 37, 37, 37, 37] and j["pairs_total"] == 300
    assert j["config4_pairs_per_rank"] == [4080] * 8
    assert j["weak_leg_frames"] == 70
    for rk in range(8):
        assert f"RANK={rk} LOCAL_RANK={rk} WORLD_SIZE=8" in r.stderr


def test_dry_run_shares_are_the_librarys():
    """The dry run's round-robin stand-in against esfm_shard_pair_list itself (host-only) for equal-cost pairs."""
    import numpy as np
    import easysfm_amd as E
    for n_img, world in ((25, 8), (25, 2), (256, 8)):
        rows = np.full(n_img, 4096, np.int32)
        mine = [len(E.shard_pair_list(n_img, rows, r, world)) for r in range(world)]
        n_pairs = n_img * (n_img - 1) // 2
        assert mine == [len(range(r, n_pairs, world)) for r in range(world)]

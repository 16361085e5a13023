#!/usr/bin/env python3
"""bench.py -- hot-path throughput of easysfm_amd on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the all-pairs SURF-64f matcher (2-NN + Lowe ratio, bit-exact with the
oracle) over this rank's share of the image-pair list, descriptors already resident in HBM.
At every N the headline is BASELINE.json configs[1] at its metric size: 25 images x 4096 features x 64
floats, 300 pairs per step (workload "M-SURF-4k"); at N > 1 that SAME pair list is
partitioned over the ranks ("scaling": "strong": total work fixed, no data-path collective), and the
round-4 form -- per-GPU work fixed at 300 pairs, F images with F(F-1)/2 >= 300 N -- is the side leg
"weak_m_surf_4k".  The second half of the metric, bundle-adjustment LM iterations/s on 25 cameras x 30k
points (240k observations, workload "BA-25"), is measured in the same process and reported under "ba" and,
in short form, inside "roofline" and "cpu_baseline" (the objects the driver's record keeps whole).

At every N the same process also runs BASELINE.json's two multi-GPU configurations and reports them as extra objects of the
one JSON line (strong scaling: the total work is fixed, N = 1 is the single-GPU point of the curve):
  "config4"  256 images x 8192 SURF features, all 32 640 pairs, pair list partitioned over the ranks, no collective
  "config5"  BA-512: 512 cams x 300k pts x 3M obs, points sharded, one RCCL all-reduce (esfm_comm_allreduce, the library's own
             communicator) of the packed reduced camera system per LM iteration

`python bench.py --gpus N` with WORLD_SIZE unset starts its N ranks itself (child processes, before anything touches a GPU).
Rank 0 prints ONE JSON line.  `value` = image pairs matched per second over all ranks.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

# The CPU oracle (checker / cpu_baseline legs) is OpenMP code: after a parallel region libgomp's workers spin for a while by
# default, which on a 256-core host starves the HIP runtime's threads of the GPU legs that follow (seen: the SURF leg at 10 ms
# per image instead of 1.5).  Passive waiting costs the long CPU regions nothing.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, f32 in / f32 acc
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16, 32 cycles per SIMD)
PEAK_HBM_GBS = 8000.0          # HBM3E spec (6.29 TB/s measured copy)
METRIC = "image-pairs matched/s (4096 SURF feats/img) + BA LM iters/s (25 cams, 30k pts)"
N_FEATS, DIM = 4096, 64


def frames_for(world: int) -> int:
    """Images of the WEAK side leg at N > 1: the smallest F with F (F - 1) / 2 >= 300 N, so that every GPU keeps 300 pairs of
    4096 x 4096 per step (round 4's headline; its value rises ~N x by construction, which is why it is a side leg now)."""
    f = 25
    while f * (f - 1) // 2 < 300 * world:
        f += 1
    return f


def host_cpu_info():
    """What the host offers this process: logical CPUs, the affinity mask, the cgroup CPU quota (v2 cpu.max / v1 cfs quota).  The
    usable parallelism is the minimum of the three -- os.cpu_count() alone (256 on the GPU boxes) says nothing about it."""
    info = {"os_cpu_count": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        info["affinity"] = None
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, b = f.read().split()[:2]
            quota = None if a == "max" else float(a) / float(b)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except Exception:
            quota = None
    info["cgroup_cpu_quota"] = quota
    try:
        with open("/proc/cpuinfo") as f:
            info["model"] = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), None)
    except Exception:
        info["model"] = None
    usable = info["affinity"] or info["os_cpu_count"]
    if quota:
        usable = max(1, min(usable, int(quota + 0.5)))
    info["usable"] = usable
    return info


def thread_ladder(usable: int):
    """Thread counts to try: 1, 4, 8, 16, 32, 64, 128 and everything usable."""
    return sorted({t for t in (1, 4, 8, 16, 32, 64, 128) if t < usable} | {usable})


def cpu_baseline_match(sets, pairs, gpu_results=None, budget_s: float = 10.0):
    """Oracle (exact brute-force 2-NN + ratio; 8-lane SIMD body in the canonical summation order, ONE parallel region over all
    (pair, query) items, oracle/match_ref.c esfm_ref_match_pairs_l2) on the host cores, on whole 4096 x 4096 pairs of the same workload.
    The thread count is SWEPT (1, 4, 8 ... everything the affinity mask / cgroup quota allow; ~1 s each on a sample of the step's pairs)
    and the best one reported with ITS thread count as `cores` (round 4 reported os.cpu_count() = 256 "cores" at 7 x the one-thread
    rate); that setting then runs the step's whole pair list until ~budget_s, and every pair's match list is compared bit for bit
    with the GPU's when `gpu_results` is given.  Returns (cpu_baseline object, verification object)."""
    import oracle
    path = oracle.build(arch="native", out="libesfm_oracle_native.so")
    oracle.load(path)
    host = host_cpu_info()
    pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
    sweep = {}
    for t in thread_ladder(host["usable"]):
        oracle.set_num_threads(t)
        k = max(1, min(len(pairs), 2 * t, 48))
        oracle.match_pairs_l2(sets, pairs[:min(2, len(pairs))], 0.5)    # thread pool warm-up at this size
        t0 = time.perf_counter(); n = 0
        while True:
            oracle.match_pairs_l2(sets, pairs[:k], 0.5); n += k
            if time.perf_counter() - t0 >= 1.0:
                break
        sweep[t] = n / (time.perf_counter() - t0)
    best = max(sweep, key=lambda t: sweep[t])
    oracle.set_num_threads(best)
    n, el, first = 0, 0.0, None
    while True:
        t0 = time.perf_counter()
        res = oracle.match_pairs_l2(sets, pairs, 0.5)
        el += time.perf_counter() - t0
        n += len(pairs)
        if first is None:
            first = res
        if el >= budget_s or n >= 64 * len(pairs):
            break
    checked, bad = 0, []
    if gpu_results is not None:
        for k, ((q, t, d), (rq, rt, rd)) in enumerate(zip(gpu_results, first)):
            if not (np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(d.view(np.uint32), rd.view(np.uint32))):
                bad.append(k)
            checked += 1
    oracle.set_num_threads(host["usable"])
    base = {"value": n / el, "unit": "image-pairs/s", "cores": best, "kind": "port",
            "sample": f"{n} pairs of 4096x4096x64 (M-SURF-4k: the step's pair list, {n // len(pairs)} times) in {el:.1f}s on {best} threads (the best of the "
                      "sweep); SIMD brute force (8 train rows per 256-bit register, canonical summation order), one OpenMP region over all (pair, query) items",
            "thread_sweep_pairs_per_s": {str(t): v for t, v in sweep.items()}, "host": host,
            "one_thread": {"value": sweep.get(1), "unit": "image-pairs/s", "cores": 1, "kind": "port", "sample": "~1 s of the same pairs, one thread"}}
    ver = None
    if gpu_results is not None:
        ver = {"ok": not bad and checked > 0, "pairs_checked": checked, "pairs_per_step": len(pairs), "queries_checked": checked * N_FEATS,
               "mismatching_pairs": bad[:16],
               "what": "every (queryIdx, trainIdx, distance bits) of the ratio-test survivors, match lists in order"}
    return base, ver


def cpu_baseline_ba(scene, iters: int = 25):
    """The oracle's LM loop on BA-25: at the reference's 4 threads (ceres_options_->num_threads = 4, ba.cpp:203) -- the headline of
    this object, as BASELINE.md section 3 asks -- and at the best thread count of a sweep (8 LM iterations each), reported with its
    thread count."""
    import oracle

    def run(threads, n_it):
        oracle.set_num_threads(threads)
        opt = oracle.ba_default_options()
        opt.max_num_iterations = n_it; opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
        _, _, s_ = oracle.ba_solve(scene.cam_idx, scene.pt_idx, scene.uv, scene.K4, scene.cams0, scene.pts0, opt)
        return {"value": s_.num_iterations / s_.solve_seconds, "unit": "LM iters/s", "cores": threads, "kind": "port",
                "sample": f"{s_.num_iterations} LM iterations of BA-25 (25 cams, 30k pts, 240k obs) in {s_.solve_seconds:.1f}s on {threads} threads"}
    host = host_cpu_info()
    out = run(min(4, host["usable"]), iters)
    sweep = {t: run(t, 8)["value"] for t in thread_ladder(host["usable"]) if t <= 64 or t == host["usable"]}
    best = max(sweep, key=lambda t: sweep[t])
    out["best_of_sweep"] = run(best, iters)
    out["thread_sweep_lm_iters_per_s"] = {str(t): v for t, v in sweep.items()}
    oracle.set_num_threads(host["usable"])
    return out


def _write_png_rgb(path: str, rgb: np.ndarray) -> None:
    """8-bit RGB PNG, filter 0, one IDAT (zlib): enough for the drivers' own reader, no imaging library needed on the box."""
    import struct
    import zlib
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 1)) + chunk(b"IEND", b""))


def e2e_leg(tag: str, feature: str, param: int, with_cpu: bool):
    """BASELINE configs 1 / 3 end to end: ./bin/sfm_native (C++ host over the C ABI, batched pair loop) on the reference's 11 fountain
    images at 768 x 512 through this repo's script/run_fountain_small.sh -- wall seconds and the driver's own stage split (the
    reference's only instrumentation is its per-stage `... cost = ... seconds` lines, feature_matching.cpp:141, ba.cpp:285) -- and,
    as the stated CPU baseline, the oracle's detect / match / verify stages on the same images (the stages whose inputs are a
    function of the images alone)."""
    import shutil
    import subprocess
    import tempfile
    gold = os.path.join(ROOT, "tests", "golden", "fountain11_gray.npz")
    exe = os.path.join(ROOT, "bin", "sfm_native")
    if not (os.path.exists(gold) and os.path.exists(exe)):
        return {"error": "fixture or bin/sfm_native missing"}
    imgs = np.load(gold)["images"]
    tmp = tempfile.mkdtemp(prefix="esfm_e2e_")
    try:
        data = os.path.join(tmp, "test_data")
        os.makedirs(os.path.join(data, "images_25")); os.makedirs(os.path.join(data, "k_25"))
        names = []
        for i, im in enumerate(imgs):
            names.append(f"{i:04d}.png")
            _write_png_rgb(os.path.join(data, "images_25", names[-1]), np.ascontiguousarray(np.stack([im] * 3, axis=2)))
        open(os.path.join(data, "image_list.txt"), "w").write("\n".join(names) + "\n")
        open(os.path.join(data, "k_25", "K.txt"), "w").write("689.87 0 380.17\r\n0 691.04 251.70\r\n0 0 1")
        out_ply = os.path.join(tmp, "out", "cloud.ply")
        env = dict(os.environ, SFM_DATA=data, SFM_OUT=out_ply, SFM_BIN="bin/sfm_native", FEATURE=feature, FEATURE_PARAM=str(param))
        env.pop("ESFM_PAIR_BY_PAIR", None)
        runs = []
        for _ in range(2):                    # the first run pays the process' one-off costs (library load, code objects, allocations)
            t0 = time.perf_counter()
            r = subprocess.run(["bash", os.path.join("script", "run_fountain_small.sh")], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=600)
            wall = time.perf_counter() - t0
            line = [l for l in r.stdout.splitlines() if l.startswith("stage seconds:")]
            if r.returncode != 1 or not line:
                return {"error": f"driver exit status {r.returncode}", "tail": r.stdout[-400:]}
            tok = line[-1].split()[2:]
            st, k = {}, 0
            while k + 1 < len(tok):
                try:
                    st[tok[k]] = float(tok[k + 1])
                except ValueError:
                    pass
                k += 2
            runs.append((wall, st, r.stdout))
        wall, st, log = min(runs, key=lambda x: x[0])
        n_pts = sum(1 for l in open(out_ply) if l[:1].isdigit() or l[:1] == "-") if os.path.exists(out_ply) else 0
        leg = {"metric": "seconds, 11 fountain images (768 x 512) -> sparse cloud, " + ("SURF minHessian 300" if feature == "S" else f"ORB {param} features"),
               "value": wall, "unit": "s", "higher_is_better": False, "driver": "bin/sfm_native via script/run_fountain_small.sh (13 positional arguments)",
               "first_run_wall_s": runs[0][0], "stage_seconds": st, "frames_registered": log.count("Progress: ["), "points_in_ply": n_pts,
               "verified_pairs": log.count("verified matches"), "includes": "process start, library load, PNG decode, every host<->device copy, .ply write"}
        if with_cpu:
            import oracle
            oracle.set_num_threads(host_cpu_info()["usable"])
            t0 = time.perf_counter()
            feats = [oracle.surf(im, float(param)) if feature == "S" else oracle.orb(im, int(param)) for im in imgs]
            t_det = time.perf_counter() - t0
            pairs = np.array([(i, j) for i in range(len(imgs)) for j in range(i)], np.int32)
            t0 = time.perf_counter()
            if feature == "S":
                res = oracle.match_pairs_l2([f[1] for f in feats], pairs, 0.5)
            else:
                res = [oracle.match_hamming(feats[i][1], feats[j][1], 0.8) for i, j in pairs]
            t_match = time.perf_counter() - t0
            K4 = np.array([689.87, 380.17, 691.04, 251.70], np.float32)
            t0 = time.perf_counter()
            n_ver = 0
            for (i, j), (q, t, d) in zip(pairs, res):
                if len(q) > 20:
                    a = np.ascontiguousarray(feats[i][0][q, :2], np.float32); b = np.ascontiguousarray(feats[j][0][t, :2], np.float32)
                    ok_, Er_, mr_, _it, _c = oracle.find_essential_ransac(a, b, K4, 0.99, 1.0)
                    if ok_:
                        oracle.recover_pose(Er_, a, b, K4, mr_)
                    n_ver += 1
            t_ver = time.perf_counter() - t0
            leg["cpu_baseline"] = {"kind": "port", "cores": oracle.num_threads(), "unit": "s",
                                   "stage_seconds": {"detect": t_det, "match": t_match, "verify": t_ver},
                                   "value": t_det + t_match + t_ver,
                                   "sample": f"oracle detect (sequential per image) + match (all 55 pairs, all cores) + 5-point RANSAC / recoverPose of the {n_ver} pairs "
                                             "with > 20 matches (sequential); BA, PnP and SOR are not in it (their inputs depend on the pipeline's state)",
                                   "gpu_same_stages_s": st.get("detect", 0.0) + st.get("match", 0.0) + st.get("verify", 0.0)}
        return leg
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def dry_run(args) -> int:
    """The launcher path without a GPU: every rank reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* exactly as the real run does,
    the ranks meet on MASTER_ADDR:MASTER_PORT (rank 0 listens, the others report in), and rank 0 prints the one JSON line."""
    import socket
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500"))
    # (MASTER_PORT itself may be taken: torch.distributed.run's agent keeps its store there.  The dry run meets next door.)
    port = port + 1 if port < 65535 else port - 1
    print(f"[bench dry-run] RANK={rank} LOCAL_RANK={local_rank} WORLD_SIZE={world} MASTER={addr}:{port}", file=sys.stderr, flush=True)
    n_pairs = 300                                # the headline: M-SURF-4k's pair list partitioned over the ranks
    # esfm_shard_pair_list on equal-cost pairs is a round robin (greedy least-loaded rank, ties to the lower rank): rank r takes
    # the pairs r, r + world, ... of the (i, j < i) list -- computed here without loading the library (no torch / HIP in a dry run)
    mine = len(range(rank, n_pairs, world))
    seen = {rank: mine}
    if world > 1:
        if rank == 0:
            srv = socket.socket(); srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port)); srv.listen(world); srv.settimeout(60.0)
            for _ in range(world - 1):
                c, _a = srv.accept()
                r, m = (int(x) for x in c.makefile().readline().split())
                seen[r] = m; c.close()
            srv.close()
        else:
            deadline = time.time() + 60.0
            while True:
                try:
                    c = socket.create_connection((addr, port), timeout=5.0); break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            c.sendall(f"{rank} {mine}\n".encode()); c.close()
    if rank == 0:
        ok = sorted(seen) == list(range(world)) and sum(seen.values()) == n_pairs
        print(json.dumps({"metric": METRIC, "value": None, "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "dry_run": True, "ranks_seen": sorted(seen), "pairs_per_rank": [seen[r] for r in sorted(seen)],
                          "pairs_total": n_pairs, "scaling": "strong",
                          "config4_pairs_per_rank": [len(range(r, 256 * 255 // 2, world)) for r in range(world)],
                          "weak_leg_frames": frames_for(world), "torch_imported": "torch" in sys.modules, "ok": bool(ok)}), flush=True)
        return 0 if ok else 4
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # 0.3 s of timed work: the first tens of steps run ~3 % slower (clock ramp), 20 were noisy
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ba-iters", type=int, default=50, help="LM iterations timed for the BA half of the metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-config45", action="store_true", help="skip the config-4 / config-5 strong-scaling legs")
    ap.add_argument("--no-e2e", action="store_true", help="skip the config-1 / config-3 end-to-end legs (bin/sfm_native on the fountain images)")
    ap.add_argument("--config4-steps", type=int, default=2)
    ap.add_argument("--ba512-iters", type=int, default=20)
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise the launcher and the rank environment only: no torch / HIP import, no GPU; ranks meet over a "
                         "TCP socket on MASTER_ADDR:MASTER_PORT and rank 0 prints one JSON line with n_gpus")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand without a launcher: become the launcher.  Children are ordinary processes started BEFORE this one has
        # touched a GPU (no torch import, no HIP call so far); rank 0 inherits stdout and prints the JSON line.
        import socket
        import subprocess
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rcs = [p.wait() for p in procs]
        return max(rcs, key=abs)

    if args.dry_run:
        return dry_run(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        print("[bench] no GPU visible: easysfm_amd has no CPU fallback", file=sys.stderr)
        return 2
    # ESFM_BENCH_BACKEND=gloo: REHEARSAL of the multi-rank run on fewer GPUs than ranks (the builder's box has one): the N ordinary
    # processes share the visible devices round-robin, torch's group is gloo, the sharded BA legs exchange through the callback over that
    # group (RCCL refuses two ranks on one device).  Everything else -- the launcher, the shards, the agreement on the communicator, the
    # legs, rank 0's JSON assembly -- is the code the real N-GPU run executes.  The line says `"backend": "gloo-rehearsal"`; its
    # throughput figures are N processes time-sharing a GPU and mean nothing.
    rehearsal = os.environ.get("ESFM_BENCH_BACKEND", "nccl") == "gloo"
    n_dev = torch.cuda.device_count()
    if rehearsal:
        local_rank = local_rank % max(n_dev, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import easysfm_amd as E
    from easysfm_amd import _lib, synth

    # ---------------------------------------------------------------- matching workload
    n_frames = 25                                  # M-SURF-4k at every N (N > 1: its 300 pairs partitioned over the ranks)
    sets = synth.surf_like_sets(n_frames, N_FEATS, pool=16384, seed_base=1000)
    rows = np.full(n_frames, N_FEATS, np.int32)
    pairs = E.shard_pair_list(n_frames, rows, rank, world)
    bank = E.DescriptorBank(sets, E.ESFM_L2_F32, device=f"cuda:{local_rank}")
    pm = E.PairMatcher(bank, pairs)
    ctx = pm.ctx
    ratio = 0.5  # matchFeaturesSURF default (feature_matching.h:21)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        pm.match(ratio)
    barrier()
    ctx.set_kernel_timing(True)
    ctx.kernel_time(_lib.K_L2_KNN); ctx.kernel_time(_lib.K_L2_RESCAN); ctx.kernel_time(_lib.K_L2_SECOND)   # drain
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pm.match(ratio)
    ctx.synchronize()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    k_ms, k_n = ctx.kernel_time(_lib.K_L2_KNN)
    s_ms, s_n = ctx.kernel_time(_lib.K_L2_SECOND)
    ctx.set_kernel_timing(False)
    n_q, n_rescan = pm.stats()
    n_second = pm.second_pass()
    # the per-bank cost that sits OUTSIDE the step: the operand images / norms the matcher derives once per resident descriptor bank
    # (PairMatcher.prepare -> esfm_match_prepare_dev: l2_split_bf16_kernel + l2_blockmax_kernel)
    ctx.synchronize()
    t0p = time.perf_counter()
    for _ in range(20):
        pm.prepare()
    ctx.synchronize()
    prepare_ms = (time.perf_counter() - t0p) / 20 * 1e3

    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    npairs = torch.tensor([float(len(pairs))], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(npairs, op=dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    total_pairs = float(npairs.item())
    value = total_pairs * args.steps / elapsed

    # correctness check (outside the timed region).  With the CPU baseline leg on (rank 0, N = 1) the oracle runs the step's WHOLE
    # pair list there and every pair's match list is compared bit for bit (`verified_scope` says how much was checked); otherwise
    # a spot check: 512 queries of this rank's first pair.
    verified, verified_scope, res_host = None, None, None
    if rank == 0:
        try:
            import oracle
            res_host = pm.match(ratio).to_host()
            i, j = pairs[0]
            rq, rt, rd = oracle.match_l2(sets[i][:512], sets[j], ratio)
            q, t, d = res_host[0]
            m = q < 512
            verified = bool(np.array_equal(q[m], rq) and np.array_equal(t[m], rt) and
                            np.array_equal(d[m].view(np.uint32), rd.view(np.uint32)))
            verified_scope = "spot check: 512 queries of pair 0"
        except Exception as e:  # the checker must never take the measurement down
            verified = f"check failed to run: {e!r}"

    flops_per_launch = 2.0 * float(len(pairs)) * N_FEATS * N_FEATS * DIM
    avg_kernel_s = (k_ms / max(k_n, 1)) * 1e-3
    achieved = flops_per_launch / avg_kernel_s / 1e12 if avg_kernel_s > 0 else 0.0
    # HBM/fabric bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate runs of this command, gfx950 correction applied; see profiles/traffic.json)
    traffic = traffic_ba = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            # (the committed figure is the 300-pair launch's: at N > 1 a rank launches its share of the list, for which no counter pass exists)
            traffic = tj.get("l2_knn_bf16x1_kernel_bytes_per_launch", tj.get("l2_knn_bf16_kernel_bytes_per_launch")) if world == 1 else None
            traffic_ba = {"ba": tj.get("ba_linearize_kernel_bytes_per_launch"), "config5": tj.get("ba512_linearize_kernel_bytes_per_launch")}
        except Exception:
            traffic = traffic_ba = None
    # The distance pass runs on the bf16 matrix cores, so the kernel is priced against the dense bf16 MFMA peak; `achieved` is
    # ALGORITHMIC (2 Nq Nt 64 per pair).  Round 3: ONE bf16 product per f32 product (l2_knn_bf16x1_kernel; the operand
    # rounding is inside the certificate's bound), so the executed MFMA work equals the algorithmic work plus the threshold-filter
    # pass over the queries the first pass leaves uncertified AND undecided (l2_finish_kernel: `second_pass_queries_per_step` x Nt x 64 x 2;
    # since round 4's ratio screen that is a handful per step).
    exec_flops = flops_per_launch + 2.0 * float(n_second) * N_FEATS * DIM
    roofline = {"bound": "mfma", "kernel": "l2_knn_bf16x1_kernel", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_MFMA_TFLOPS, "traffic": traffic,
                "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not measured in this run)" if traffic else None,
                "avg_launch_ms": avg_kernel_s * 1e3, "launches": k_n, "pairs_per_launch_rank0": int(len(pairs)),
                "algorithmic_flops_per_launch": flops_per_launch,
                "executed_mfma_flops_per_step": exec_flops, "products_per_f32_product": 1,
                "f32_mfma_peak": PEAK_F32_MFMA_TFLOPS, "achieved_over_f32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                "finish_kernel": "l2_finish_kernel (threshold-filter second pass + brute force of overflowed chunks + ratio test + compaction)",
                "finish_kernel_avg_ms": (s_ms / max(s_n, 1)), "second_pass_queries_per_step": n_second,
                "launches_per_step": 2, "step_minus_kernels_ms": elapsed / args.steps * 1e3 - (k_ms / max(k_n, 1)) - (s_ms / max(s_n, 1)),
                "rescanned_queries_per_step": n_rescan, "queries_per_step": n_q,
                "prepare_ms": prepare_ms,
                "prepare_note": "once per resident descriptor bank (bf16 operand images, norms), outside the timed step; value_including_prepare charges it to EVERY step",
                "value_including_prepare": total_pairs / (elapsed / args.steps + prepare_ms * 1e-3)}

    out = {
        "metric": METRIC, "value": value, "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "scaling_note": "total work fixed: M-SURF-4k's 300 pairs of 4096 x 4096 per step at every N, the pair list partitioned over the ranks "
                        "(37-38 pairs per GPU at N = 8: a 0.07 ms step per rank); per-GPU work fixed at 300 pairs: side leg weak_m_surf_4k; "
                        "BASELINE's multi-GPU configurations: legs config4 / config5 (strong)",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "M-SURF-4k all-pairs SURF-64f match (2-NN + ratio 0.5), "
                               f"{n_frames} imgs x {N_FEATS} feats x {DIM} f32, {int(total_pairs)} pairs/step, "
                               "pair list partitioned over ranks, no collective",
                   "pairs_per_step": int(total_pairs), "features_per_image": N_FEATS, "descriptor_dim": DIM, "ratio": ratio},
        "roofline": roofline, "verified_vs_oracle": verified, "verified_scope": verified_scope,
    }
    if world > 1:
        out["backend"] = "gloo-rehearsal" if rehearsal else "nccl"
        out["gpus_physical"] = n_dev
        out["pairs_per_rank"] = None          # (filled below)
        pr_ = [None] * world
        dist.all_gather_object(pr_, int(len(pairs)))
        out["pairs_per_rank"] = pr_

    roofline["parity"] = {"verified_vs_oracle": verified, "verified_scope": verified_scope}
    roofline["legs"] = {}
    if world > 1:
        # round 4's headline as a side leg: per-GPU work fixed at 300 pairs per step (F images, F (F - 1) / 2 >= 300 N pairs over the ranks)
        try:
            n_fw = frames_for(world)
            sets_w = synth.surf_like_sets(n_fw, N_FEATS, pool=16384, seed_base=1000)
            pairs_w = E.shard_pair_list(n_fw, np.full(n_fw, N_FEATS, np.int32), rank, world)
            pm_w = E.PairMatcher(E.DescriptorBank(sets_w, E.ESFM_L2_F32, device=f"cuda:{local_rank}"), pairs_w)
            for _ in range(args.warmup):
                pm_w.match(ratio)
            pm_w.ctx.synchronize(); torch.cuda.synchronize(dev); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                pm_w.match(ratio)
            pm_w.ctx.synchronize(); torch.cuda.synchronize(dev); dist.barrier()
            ts = torch.tensor([time.perf_counter() - t0, float(len(pairs_w))], dtype=torch.float64, device=dev)
            tmax = ts[:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            psum = ts[1:].clone(); dist.all_reduce(psum, op=dist.ReduceOp.SUM)
            out["weak_m_surf_4k"] = {"value": float(psum.item()) * args.steps / float(tmax.item()), "unit": "image-pairs/s", "n_gpus": world, "scaling": "weak",
                                     "ms_per_step": float(tmax.item()) / args.steps * 1e3, "pairs_this_rank": int(len(pairs_w)),
                                     "config": {"workload": f"{n_fw} imgs x 4096 feats, {int(psum.item())} pairs per step over the ranks (300 per GPU)"}}
            roofline["legs"]["weak_m_surf_4k"] = {"value": out["weak_m_surf_4k"]["value"], "unit": "image-pairs/s", "pairs_per_step": int(psum.item())}
            pm_w.close()
            del pm_w
        except Exception as e:
            out["weak_m_surf_4k"] = {"error": repr(e)}

    # ---------------------------------------------------------------- BA half of the metric
    printed = threading.Event()

    def finalize():
        """The line as the driver's record keeps it (BENCH_rNN.json `parsed`): of `roofline` the first ~24 scalar keys, of `cpu_baseline`
        five, strings cut at ~128 characters, nested objects dropped; of stdout the last 8 KB.  So: both halves of the metric, the
        parity flags and the side legs' headline figures go FIRST in `roofline` as flat scalars (round 5's record lost the BA half for
        that), descriptive strings last; `cpu_baseline` leads with the four required keys and the BA half's figures; and the `ba`
        object is the LAST key of the line so the stdout tail holds it whole."""
        def g(obj, *path):
            for k in path:
                if not isinstance(obj, dict) or k not in obj:
                    return None
                obj = obj[k]
            return obj
        r = out["roofline"]
        ba, c4, c5, hard, orb, c2 = out.get("ba"), out.get("config4"), out.get("config5"), out.get("hard"), out.get("orb"), out.get("config2")
        head = {"bound": r["bound"], "kernel": r["kernel"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"], "traffic": r["traffic"],
                "verified_vs_oracle": out.get("verified_vs_oracle"), "ba_verified_vs_oracle": g(ba, "verified_vs_oracle"),
                "ba_lm_iters_per_s": g(ba, "value"), "ba_ms_per_iteration": g(ba, "ms_per_iteration"),
                "ba_sweep_gbps": g(ba, "roofline", "achieved"), "ba_sweep_frac": g(ba, "roofline", "frac"),
                "value_including_prepare": r.get("value_including_prepare"),
                "config4_pairs_per_s": g(c4, "value"), "config4_frac": g(c4, "roofline_rank0", "frac"), "config5_lm_iters_per_s": g(c5, "value"),
                "config5_sweep_frac": g(c5, "roofline", "frac"),
                "hard_pairs_per_s": g(hard, "value"), "hard_vs_benign": g(hard, "vs_benign"), "orb_pairs_per_s": g(orb, "value"),
                "config2_pairs_per_s": g(c2, "value"), "config2_verified_vs_oracle": g(c2, "verified_vs_oracle"),
                "rccl_ranks": out.get("rccl_ranks", 1 if world == 1 else None)}
        out["roofline"] = {**head, **{k: v for k, v in r.items() if k not in head}}
        cb = out.get("cpu_baseline")
        if isinstance(cb, dict) and "error" not in cb:
            lead = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                    "ba_lm_iters_per_s_4_threads": g(cb, "ba", "value"), "ba_lm_iters_per_s_best": g(cb, "ba", "best_of_sweep", "value"),
                    "ba_best_cores": g(cb, "ba", "best_of_sweep", "cores"), "one_thread_pairs_per_s": g(cb, "one_thread", "value")}
            out["cpu_baseline"] = {**lead, **{k: v for k, v in cb.items() if k not in lead}}
        first = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "verified_vs_oracle", "config", "roofline", "cpu_baseline"]
        last = ["config4", "config5", "hard", "config2", "ba"]
        ordered = {k: out[k] for k in first if k in out}
        ordered.update({k: v for k, v in out.items() if k not in first and k not in last})
        ordered.update({k: out[k] for k in last if k in out})
        return ordered

    def emit():
        if rank == 0 and not printed.is_set():
            printed.set()
            try:
                line = finalize()
            except Exception as e:      # (the line must come out whatever the ordering step meets)
                line = dict(out, finalize_error=repr(e))
            print(json.dumps(line), flush=True)

    # the library's own RCCL communicator for the sharded BA legs (esfm_comm_*: no callback into Python per all-reduce);
    # the 128-byte id travels through torch.distributed's store
    bctx = E.Context.on_torch_stream(local_rank)
    comm = None
    if world > 1:
        # every step may fail on a node this code has never seen (no round had more than one GPU): the ranks AGREE on the outcome
        # (MIN over an ok flag through torch's process group) and, if the library's communicator is not there on all of them, the
        # sharded BA legs go through torch.distributed's own RCCL group instead (easysfm_amd.ba.torch_allreduce_callback) -- the
        # line says which (`ba_allreduce_via`).  The matching legs need no communicator at all.
        # ncclCommInitRank is itself a collective: a rank that fails BEFORE joining (library not loadable, no id, a bad device) would
        # leave the others waiting inside it for ever, so (1) everything that can be checked beforehand is checked and AGREED on first,
        # and nobody calls esfm_comm_create unless every rank passed; (2) the create runs in a helper thread with a deadline -- a rank
        # that does not come back reports failure at the second agreement instead of hanging the run (its thread is left behind: daemon).
        # ESFM_BENCH_FAIL_COMM_RANK=k forces rank k's pre-check to fail (the rehearsal test's lever).
        def agree(ok: bool) -> bool:
            f = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            return float(f.item()) >= 1.0

        def gather_errors(e):
            box_ = [None] * world
            dist.all_gather_object(box_, e)
            return "; ".join(f"rank {r_}: {m_}" for r_, m_ in enumerate(box_) if m_) or None

        uid, err = None, None
        try:
            uid = E.Comm.unique_id() if rank == 0 else None
        except Exception as e:
            err = repr(e)
        box = [uid]
        dist.broadcast_object_list(box, src=0)
        if box[0] is None and err is None:
            err = "no RCCL unique id from rank 0"
        if os.environ.get("ESFM_BENCH_FAIL_COMM_RANK", "") == str(rank):
            err = "forced failure (ESFM_BENCH_FAIL_COMM_RANK)"
        pre_ok = agree(err is None)                       # (every branch below is taken by all ranks or by none)
        shared = world > n_dev
        created = False
        if pre_ok and shared:
            err = f"not attempted: {world} ranks share {n_dev} device(s), RCCL refuses two ranks on one device"
        elif pre_ok:
            res_ = {}

            def create():
                try:
                    res_["comm"] = E.Comm(bctx, box[0], rank, world)
                except Exception as e:
                    res_["err"] = repr(e)
            th = threading.Thread(target=create, daemon=True)
            th.start()
            th.join(float(os.environ.get("ESFM_BENCH_COMM_TIMEOUT", "120")))
            if th.is_alive():
                err = "esfm_comm_create did not return within its deadline"
            else:
                comm, err = res_.get("comm"), res_.get("err")
            created = agree(comm is not None)
        why = gather_errors(err)
        if not created:
            if comm is not None:
                comm = None                              # (left to process teardown: a collective destroy could wait for the failed ranks)
            from easysfm_amd.ba import torch_allreduce_callback
            comm = torch_allreduce_callback()
            out["rccl_ranks"] = dist.get_world_size() if not rehearsal else 0
            out["ba_allreduce_via"] = f"torch.distributed {'gloo' if rehearsal else 'nccl'} group (esfm_comm: {why})"
        else:
            out["rccl_ranks"] = comm.rccl_ranks()        # ncclCommCount of the library's own communicator: did RCCL see N ranks
            out["ba_allreduce_via"] = "esfm_comm (library's own RCCL communicator)"
        roofline["rccl_ranks"] = out["rccl_ranks"]
        roofline["ba_allreduce_via"] = out["ba_allreduce_via"]

    def ba_leg(scene, iters, name, workload):
        """LM iterations/s on `scene`, points (hence observations) sharded over the ranks; returns the leg's JSON object."""
        if world > 1:
            shard = E.shard_points(scene.n_pt, scene.pt_idx, world)
            keep = shard[scene.pt_idx] == rank
            ci, pi, uv = scene.cam_idx[keep], scene.pt_idx[keep], scene.uv[keep]
        else:
            ci, pi, uv = scene.cam_idx, scene.pt_idx, scene.uv
        with torch.cuda.stream(bctx.torch_stream):
            prob = E.BAProblem(ci, pi, uv, scene.K4, scene.cams0, scene.pts0, bctx)
            opt = E.default_options()
            opt.function_tolerance = 0.0; opt.parameter_tolerance = 0.0; opt.gradient_tolerance = 0.0
            opt.max_num_iterations = 3
            prob.solve(opt, comm)                   # warm-up (allocations, code objects, RCCL channels)
            prob.set_params(scene.cams0, scene.pts0)
            opt.max_num_iterations = iters
            barrier()
            tb = time.perf_counter()
            summ = prob.solve(opt, comm)
            bctx.synchronize()
            if world > 1:
                dist.barrier()
            ba_el = time.perf_counter() - tb
            # the same solve once more with the library's per-kernel event timers on, for the roofline object (the events cost
            # ~25 us per LM iteration, a tenth of it on BA-25: they stay out of the timed solve)
            prob.set_params(scene.cams0, scene.pts0)
            bctx.set_kernel_timing(True)
            bctx.kernel_time(_lib.K_BA_LINEARIZE); bctx.kernel_time(_lib.K_BA_SCHUR); bctx.kernel_time(_lib.K_BA_SOLVE)
            prob.solve(opt, comm)
            bctx.synchronize()
            l_ms, l_n = bctx.kernel_time(_lib.K_BA_LINEARIZE)
            s_ms, s_n = bctx.kernel_time(_lib.K_BA_SCHUR)
            c_ms, c_n = bctx.kernel_time(_lib.K_BA_SOLVE)
            bctx.set_kernel_timing(False)
            cams_out, pts_out = prob.get_params()
            prob.close()
        tt2 = torch.tensor([ba_el], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt2, op=dist.ReduceOp.MAX)
        ba_el = float(tt2.item())
        # compulsory bytes of the Jacobian sweep = 176 Nobs + 48 Nc + 24 Np (16 B in, J 144 B + r 16 B out per
        # observation) -- the figure `achieved` / `frac` use, and since round 3 also what the kernel is built to move.
        sweep_bytes = 176.0 * len(ci) + 48.0 * scene.n_cam + 24.0 * scene.n_pt
        sweep_bytes_design = float(E.ba_sweep_bytes_per_obs()) * len(ci) + 48.0 * scene.n_cam + 24.0 * scene.n_pt
        lin_s = (l_ms / max(l_n, 1)) * 1e-3
        n_red = 6 * scene.n_cam
        plan = E.reduced_plan(scene.n_cam, scene.n_pt, scene.cam_idx, scene.pt_idx)        # (host-only; the whole observation list, as the ranks' union)
        sparse = bool(plan["worthwhile"]) and os.environ.get("ESFM_BA_SOLVE", "")[:1] != "d"
        exch_mb = ((36 * plan["covisible_blocks"] + n_red) if sparse else (18 * scene.n_cam * (scene.n_cam + 1) + n_red)) * 8 / 1e6
        leg = {
            "metric": "BA LM iters/s", "value": summ.num_iterations / ba_el, "unit": "LM iters/s",
            "lm_iterations": summ.num_iterations, "seconds": ba_el, "ms_per_iteration": ba_el / max(summ.num_iterations, 1) * 1e3,
            "n_gpus": world, "scaling": "strong",
            "config": {"workload": workload, "obs_sharded_by_point": world > 1,
                       "allreduce": (f"RCCL ({'esfm_comm_allreduce' if isinstance(comm, E.Comm) else 'torch.distributed all_reduce'}) sum of the packed reduced camera system"
                                     f"{' (co-visible camera blocks only)' if sparse else ''}, {exch_mb:.2f} MB per LM iteration") if world > 1 else None},
            "reduced_solve": ({"kind": "structure-aware (nested dissection, tiles of the symbolic fill)", "tile_columns": plan["nb"], "tiles": len(plan["tiles"]),
                               "dependency_chain_tile_columns": plan["chain"], "dense_tile_columns": plan["dense_nb"],
                               "dense_tiles": plan["dense_nb"] * (plan["dense_nb"] + 1) // 2 + plan["dense_nb"], "exchange_mb_if_sharded": exch_mb}
                              if sparse else {"kind": "dense", "tile_columns": plan["dense_nb"], "exchange_mb_if_sharded": exch_mb}),
            "initial_cost": summ.initial_cost, "final_cost": summ.final_cost, "cost_trace": [it_.cost for it_ in summ.log()],
            "successful_steps": summ.num_successful_steps, "unsuccessful_steps": summ.num_unsuccessful_steps,
            "roofline": {"bound": "hbm", "kernel": "ba_linearize_kernel", "achieved": sweep_bytes / lin_s / 1e9 if lin_s > 0 else 0.0,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": (sweep_bytes / lin_s / 1e9 / PEAK_HBM_GBS) if lin_s > 0 else 0.0,
                         "traffic": (traffic_ba or {}).get(name),
                         "avg_launch_ms": lin_s * 1e3, "launches": l_n, "algorithmic_bytes_per_launch": sweep_bytes,
                         "algorithmic_bytes_formula": "176*Nobs + 48*Nc + 24*Np (SURVEY 8d)",
                         "designed_bytes_per_launch": sweep_bytes_design,
                         "frac_designed_bytes": (sweep_bytes_design / lin_s / 1e9 / PEAK_HBM_GBS) if lin_s > 0 else 0.0},
            "schur_kernel_avg_ms": s_ms / max(s_n, 1), "solve_kernel_avg_ms": c_ms / max(c_n, 1),
            # (6 Nc)^3 / 3 over the solve's time WHATEVER solve ran: the rate a dense factorisation would need for the same time.  The
            # structure-aware solve executes `reduced_solve.tiles` of `dense_tiles` tiles (BA-512: ~249 of 1 224), so its executed rate is
            # about that fraction of this figure.
            "solve_dense_equivalent_gflops_f64": (n_red ** 3 / 3.0) / (c_ms / max(c_n, 1) * 1e-3) / 1e9 if c_n else None,
            "solve_executed_tile_fraction": (len(plan["tiles"]) / max(plan["dense_nb"] * (plan["dense_nb"] + 1) // 2 + plan["dense_nb"], 1)) if sparse else 1.0,
        }
        return leg, summ

    if not args.no_ba:
        watchdog = threading.Timer(400.0, lambda: (out.setdefault("ba", {"error": "BA leg timed out"}), emit(), os._exit(3)))
        watchdog.daemon = True
        watchdog.start()
        try:
            scene = synth.ba_scene(25, 30000, 8, radius=10.0, extent=2.0, seed=4000)
            out["ba"], _ = ba_leg(scene, args.ba_iters, "ba",
                                  "BA-25: 25 cams x 30000 pts x 240000 obs (8 obs/pt), Cauchy(0.5), DENSE_SCHUR-style LM")
            # the BA half of the metric where the driver's record keeps it: inside `roofline`
            b_ = out["ba"]
            roofline["ba"] = {"metric": "BA LM iters/s (25 cams, 30k pts, 240k obs)", "value": b_["value"], "unit": "LM iters/s", "ms_per_iteration": b_["ms_per_iteration"],
                              "lm_iterations": b_["lm_iterations"], "bound": "hbm", "kernel": "ba_linearize_kernel", "achieved": b_["roofline"]["achieved"],
                              "peak": PEAK_HBM_GBS, "unit_roofline": "GB/s", "frac": b_["roofline"]["frac"], "traffic": b_["roofline"]["traffic"],
                              "avg_launch_ms": b_["roofline"]["avg_launch_ms"], "algorithmic_bytes_per_launch": b_["roofline"]["algorithmic_bytes_per_launch"],
                              "schur_kernel_avg_ms": b_["schur_kernel_avg_ms"], "solve_kernel_avg_ms": b_["solve_kernel_avg_ms"]}
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                # parity at the metric's size: the first LM iterations against the oracle (cost 1e-9, accept pattern exact)
                try:
                    import oracle
                    o5 = E.default_options(); o5.max_num_iterations = 5
                    _c, _p, g5 = E.ba_solve(scene.cam_idx, scene.pt_idx, scene.uv, scene.K4, scene.cams0, scene.pts0, o5, bctx)
                    r5 = oracle.ba_default_options(); r5.max_num_iterations = 5
                    oracle.set_num_threads(min(16, os.cpu_count() or 1))
                    _rc, _rp, rs5 = oracle.ba_solve(scene.cam_idx, scene.pt_idx, scene.uv, scene.K4, scene.cams0, scene.pts0, r5)
                    oracle.set_num_threads(os.cpu_count() or 1)
                    ok = g5.num_iterations == rs5.num_iterations
                    for a_, b_ in zip(g5.log(), oracle.iterations(rs5)):
                        ok = ok and a_.step_is_successful == b_.step_is_successful and abs(a_.cost - b_.cost) <= 1e-9 * max(1.0, abs(b_.cost))
                    out["ba"]["verified_vs_oracle"] = bool(ok and np.allclose(_c, _rc, rtol=1e-6, atol=1e-6) and np.allclose(_p, _rp, rtol=1e-6, atol=1e-6))
                except Exception as e:
                    out["ba"]["verified_vs_oracle"] = f"check failed to run: {e!r}"
                roofline["parity"]["ba_verified_vs_oracle"] = out["ba"]["verified_vs_oracle"]
                roofline["parity"]["ba_verified_scope"] = "5 LM iterations of BA-25 at full size: cost trace 1e-9, accept pattern, parameters 1e-6"
        except Exception as e:
            out["ba"] = {"error": repr(e)}
        watchdog.cancel()

    # ---------------------------------------------------------------- BASELINE configs 4 and 5 (strong scaling over the ranks)
    if not args.no_config45 and not args.no_ba:
        watchdog = threading.Timer(900.0, lambda: (out.setdefault("config4", {"error": "config 4/5 legs timed out"}), emit(), os._exit(3)))
        watchdog.daemon = True
        watchdog.start()
        try:
            n_img4, n_feat4 = 256, 8192
            sets4 = synth.surf_like_sets(n_img4, n_feat4, pool=65536, seed_base=2000)
            pairs4 = E.shard_pair_list(n_img4, np.full(n_img4, n_feat4, np.int32), rank, world)
            bank4 = E.DescriptorBank(sets4, E.ESFM_L2_F32, device=f"cuda:{local_rank}")
            pm4 = E.PairMatcher(bank4, pairs4)
            pm4.match(ratio)
            pm4.ctx.synchronize(); torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            pm4.ctx.set_kernel_timing(True); pm4.ctx.kernel_time(_lib.K_L2_KNN)
            t0 = time.perf_counter()
            for _ in range(args.config4_steps):
                res4 = pm4.match(ratio)
            pm4.ctx.synchronize(); torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            el4 = time.perf_counter() - t0
            k4_ms, k4_n = pm4.ctx.kernel_time(_lib.K_L2_KNN)
            pm4.ctx.set_kernel_timing(False)
            tt4 = torch.tensor([el4], dtype=torch.float64, device=dev)
            nm4 = torch.tensor([float(res4.n_out.sum().item()), float(len(pairs4)), float(pm4.stats()[1])], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(tt4, op=dist.ReduceOp.MAX); dist.all_reduce(nm4, op=dist.ReduceOp.SUM)
            el4 = float(tt4.item())
            n_pairs4 = n_img4 * (n_img4 - 1) // 2
            fl4 = 2.0 * len(pairs4) * n_feat4 * n_feat4 * DIM
            k4_s = k4_ms / max(k4_n, 1) * 1e-3
            out["config4"] = {
                "metric": "image-pairs matched/s (8192 SURF feats/img)", "value": n_pairs4 * args.config4_steps / el4, "unit": "image-pairs/s",
                "n_gpus": world, "scaling": "strong", "steps": args.config4_steps, "s_per_step": el4 / args.config4_steps,
                "config": {"workload": f"M-SURF-8k: {n_img4} imgs x {n_feat4} feats x {DIM} f32, all {n_pairs4} pairs per step, "
                                       "pair list partitioned over ranks (cost-balanced), no collective"},
                "pairs_covered": int(nm4[1].item()), "matches_per_step": int(nm4[0].item()), "rescanned_queries_per_step": int(nm4[2].item()),
                "roofline_rank0": {"bound": "mfma", "kernel": "l2_knn_bf16x1_kernel", "achieved": fl4 / k4_s / 1e12 if k4_s > 0 else 0.0,
                                   "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": fl4 / k4_s / 1e12 / PEAK_BF16_MFMA_TFLOPS if k4_s > 0 else 0.0,
                                   "avg_launch_ms": k4_s * 1e3, "pairs_this_rank": len(pairs4)},
            }
            roofline["legs"]["config4"] = {"value": out["config4"]["value"], "unit": "image-pairs/s (8192 feats/img)", "s_per_step": out["config4"]["s_per_step"],
                                           "frac_rank0": out["config4"]["roofline_rank0"]["frac"], "pairs_per_step": n_pairs4, "scaling": "strong"}
            pm4.close()
            del pm4, bank4, sets4, res4
            torch.cuda.empty_cache()
        except Exception as e:
            out["config4"] = {"error": repr(e)}
        try:
            scene5 = synth.ba_scene(512, 300000, 10, radius=40.0, extent=8.0, seed=5000)
            out["config5"], _ = ba_leg(scene5, args.ba512_iters, "config5",
                                       "BA-512: 512 cams x 300000 pts x 3000000 obs (10 obs/pt), Cauchy(0.5), DENSE_SCHUR-style LM")
            c5 = out["config5"]
            roofline["legs"]["config5"] = {"value": c5["value"], "unit": "LM iters/s (512 cams, 300k pts, 3M obs)", "ms_per_iteration": c5["ms_per_iteration"],
                                           "sweep_frac_hbm": c5["roofline"]["frac"], "schur_kernel_avg_ms": c5["schur_kernel_avg_ms"],
                                           "solve_kernel_avg_ms": c5["solve_kernel_avg_ms"], "reduced_solve": c5["reduced_solve"]["kind"],
                                           "exchange_mb_if_sharded": c5["reduced_solve"]["exchange_mb_if_sharded"], "scaling": "strong"}
        except Exception as e:
            out["config5"] = {"error": repr(e)}
        watchdog.cancel()

    # ---------------------------------------------------------------- ORB leg (row a-2): M-ORB-4k, rank 0 at N = 1
    if rank == 0 and world == 1 and not args.no_ba:
        try:
            osets = synth.orb_like_sets(25, N_FEATS, pool=16384, seed_base=3000)
            opairs = synth.all_pairs(25)
            obank = E.DescriptorBank(osets, E.ESFM_HAMMING, device=f"cuda:{local_rank}")
            opm = E.PairMatcher(obank, opairs)
            octx = opm.ctx
            for _ in range(max(args.warmup, 2) + 20):      # (the kernel settles over its first few dozen launches: 0.51 -> 0.44 ms, the clock; the headline has 5 + its steps)
                opm.match(0.8)
            octx.synchronize(); octx.set_kernel_timing(True); octx.kernel_time(_lib.K_HAMMING_KNN)
            t0 = time.perf_counter()
            n_rep = max(20, args.steps // 4)
            for _ in range(n_rep):
                opm.match(0.8)
            octx.synchronize()
            o_el = time.perf_counter() - t0
            h_ms, h_n = octx.kernel_time(_lib.K_HAMMING_KNN)
            octx.set_kernel_timing(False)
            h_s = h_ms / max(h_n, 1) * 1e-3
            ops = 2.0 * len(opairs) * N_FEATS * N_FEATS * 256          # multiply-adds x 2 of the element-per-bit formulation
            fp4 = os.environ.get("ESFM_HM_PASS", "") != "i8"            # round 4: nibble-per-bit operands on the FP4 matrix instruction (exact: products 0 / -2)
            out["orb"] = {"metric": "image-pairs matched/s (4096 ORB feats/img)", "value": len(opairs) * n_rep / o_el, "unit": "image-pairs/s",
                          "config": {"workload": "M-ORB-4k all-pairs ORB-256b match (2-NN + ratio 0.8), 25 imgs x 4096 feats x 32 B, 300 pairs/step"},
                          "kernel": "hamming_fp4_kernel (v_mfma_f32_32x32x64_f8f6f4, FP4 x FP4)" if fp4 else "hamming_knn_mfma_kernel (v_mfma_i32_32x32x32_i8)",
                          "avg_launch_ms": h_s * 1e3,
                          "roofline": {"bound": "mfma", "achieved": ops / h_s / 1e12, "peak": 10000.0 if fp4 else 3944.0,
                                       "unit": "TOP/s (FP4 dense, MI355X_MICROARCH.md)" if fp4 else "TOP/s (i8)",
                                       "frac": ops / h_s / 1e12 / (10000.0 if fp4 else 3944.0)},
                          "valu_equivalent_lane_ops_per_s": 2.0 * len(opairs) * N_FEATS * N_FEATS * 8 / h_s}
            if not args.no_cpu_baseline:
                import oracle
                res = opm.match(0.8).to_host()
                oracle.set_num_threads(host_cpu_info()["usable"])
                t0 = time.perf_counter()
                bad = [p_ for p_, ((i, j), (q, t, d)) in enumerate(zip(opairs, res))
                       if not all(np.array_equal(x_, y_) for x_, y_ in zip((q, t, d), oracle.match_hamming(osets[i], osets[j], 0.8)))]
                t_or = time.perf_counter() - t0
                out["orb"]["verified_vs_oracle"] = not bad
                out["orb"]["verified_scope"] = (f"all {len(opairs)} of {len(opairs)} pairs of the step ({len(opairs) * N_FEATS} queries): every (queryIdx, trainIdx, distance) "
                                                "of the ratio-test survivors, match lists in order")
                if bad:
                    out["orb"]["verified_mismatching_pairs"] = bad[:16]
                out["orb"]["cpu_baseline"] = {"value": len(opairs) / t_or, "unit": "image-pairs/s", "cores": oracle.num_threads(), "kind": "port",
                                              "sample": f"the step's {len(opairs)} pairs once in {t_or:.1f}s (popcount brute force, OpenMP over query rows; the run that checks the GPU's lists)"}
            roofline["legs"]["orb"] = {"value": out["orb"]["value"], "unit": "image-pairs/s (4096 ORB feats/img)", "frac": out["orb"]["roofline"]["frac"],
                                       "kernel": out["orb"]["kernel"], "verified_vs_oracle": out["orb"].get("verified_vs_oracle"), "verified_scope": out["orb"].get("verified_scope")}
            opm.close()
        except Exception as e:
            out["orb"] = {"error": repr(e)}

    # ---------------------------------------------------------------- M-SURF-4k-hard, rank 0 at N = 1
    # The headline's generator is benign for this matcher: its "fresh" rows are isotropic, the ratio screen drops 96 % of the queries
    # and nothing reaches the second pass.  The same shape from REAL descriptors -- the reference's fountain images' SURF-300 rows,
    # resampled (synth.msurf4k_hard_sets) -- and the headline's own sets at ratio 0.8: pairs/s, the two kernels, how many queries
    # survive the screen / reach the second pass / are re-scanned, every pair's match list against the oracle.
    if rank == 0 and world == 1 and not args.no_ba:
        try:
            gold = os.path.join(ROOT, "tests", "golden", "fountain11_gray.npz")
            hctx = E.Context(local_rank, None)
            fsets = [E.surf_detect_and_compute(im, 300.0, None, hctx)[1] for im in np.load(gold)["images"]]
            fpool = np.concatenate(fsets)
            hsets = synth.msurf4k_hard_sets(fpool)
            hpairs = synth.all_pairs(25)
            # BASELINE config 2: the 11 fountain images' own SURF-300 descriptors (768 x 512: 2 - 3 k rows per image), resident, all 55
            # (i, j < i) pairs at ratio 0.5 in one batched call
            try:
                pairs2 = synth.all_pairs(len(fsets))
                pm2 = E.PairMatcher(E.DescriptorBank(fsets, E.ESFM_L2_F32, device=f"cuda:{local_rank}"), pairs2)
                for _ in range(5):
                    pm2.match(0.5)
                pm2.ctx.synchronize()
                n_rep2 = max(20, args.steps // 4)
                t0 = time.perf_counter()
                for _ in range(n_rep2):
                    r2 = pm2.match(0.5)
                pm2.ctx.synchronize()
                el2 = (time.perf_counter() - t0) / n_rep2
                host2 = r2.to_host()
                out["config2"] = {"metric": "image-pairs matched/s (fountain, SURF minHessian 300)", "value": len(pairs2) / el2, "unit": "image-pairs/s",
                                  "ms_per_call": el2 * 1e3, "pairs": len(pairs2), "rows_per_image": [int(len(x_)) for x_ in fsets],
                                  "matches": int(sum(len(x_[0]) for x_ in host2)),
                                  "config": {"workload": "BASELINE config 2: fountain 11 imgs (768 x 512), their own SURF-64f descriptors resident, all 55 pairs, 2-NN + ratio 0.5"}}
                if not args.no_cpu_baseline:
                    import oracle
                    oracle.set_num_threads(host_cpu_info()["usable"])
                    t0 = time.perf_counter()
                    ref2 = oracle.match_pairs_l2(fsets, pairs2, 0.5)
                    t2 = time.perf_counter() - t0
                    out["config2"]["verified_vs_oracle"] = all(np.array_equal(a_[0], b_[0]) and np.array_equal(a_[1], b_[1]) and
                                                               np.array_equal(a_[2].view(np.uint32), b_[2].view(np.uint32)) for a_, b_ in zip(host2, ref2))
                    out["config2"]["verified_scope"] = "all 55 pairs: every (queryIdx, trainIdx, distance bits), match lists in order"
                    out["config2"]["cpu_baseline"] = {"value": len(pairs2) / t2, "unit": "image-pairs/s", "cores": oracle.num_threads(), "kind": "port",
                                                      "sample": f"the same 55 pairs once in {t2:.2f}s (the run that checks the GPU's lists)"}
                pm2.close()
            except Exception as e:
                out["config2"] = {"error": repr(e)}

            def match_leg(sets_, ratio_, n_rep):
                pm_ = E.PairMatcher(E.DescriptorBank(sets_, E.ESFM_L2_F32, device=f"cuda:{local_rank}"), hpairs)
                c_ = pm_.ctx
                for _ in range(3):
                    pm_.match(ratio_)
                c_.synchronize(); c_.set_kernel_timing(True); c_.kernel_time(_lib.K_L2_KNN); c_.kernel_time(_lib.K_L2_SECOND)
                t0_ = time.perf_counter()
                for _ in range(n_rep):
                    r_ = pm_.match(ratio_)
                c_.synchronize()
                el_ = (time.perf_counter() - t0_) / n_rep
                k_ = c_.kernel_time(_lib.K_L2_KNN); f_ = c_.kernel_time(_lib.K_L2_SECOND)
                c_.set_kernel_timing(False)
                nq_, nres_ = pm_.stats(); nsec_ = pm_.second_pass()
                host_ = r_.to_host()
                pm_.set_l2_audit(4); pm_.match(ratio_); c_.synchronize(); nrej_ = len(pm_.flagged()); pm_.set_l2_audit(0)
                pm_.close()
                return {"value": len(hpairs) / el_, "unit": "image-pairs/s", "ratio": ratio_, "ms_per_step": el_ * 1e3,
                        "pass_kernel_ms": k_[0] / max(k_[1], 1), "finish_kernel_ms": f_[0] / max(f_[1], 1), "queries_per_step": nq_,
                        "screen_survivors_per_step": nq_ - nrej_, "second_pass_queries_per_step": nsec_, "rescanned_queries_per_step": nres_,
                        "matches_per_step": int(sum(len(x_[0]) for x_ in host_))}, host_
            n_rep = max(10, args.steps // 4)
            legs_h, res_h = {}, {}
            for name_, sets_, ratio_ in (("hard_ratio_0.5", hsets, 0.5), ("hard_ratio_0.8", hsets, 0.8), ("m_surf_4k_ratio_0.8", sets, 0.8)):
                legs_h[name_], res_h[name_] = match_leg(sets_, ratio_, n_rep)
            out["hard"] = {"metric": "image-pairs matched/s (4096 SURF feats/img), M-SURF-4k-hard",
                           "config": {"workload": "M-SURF-4k-hard: 25 imgs x 4096 feats x 64 f32, 300 pairs/step; rows = the fountain images' own SURF-300 descriptors "
                                                  f"(first {synth.HARD_POOL_ROWS} of {len(fpool)}) resampled: half of every image re-observes a shared set of 4096 rows with "
                                                  f"N(0, {synth.HARD_TRACK_NOISE}^2) noise, half perturbs other rows with N(0, {synth.HARD_FRESH_NOISE}^2); re-normalised"},
                           **legs_h, "value": legs_h["hard_ratio_0.5"]["value"], "unit": "image-pairs/s",
                           "vs_benign": legs_h["hard_ratio_0.5"]["value"] / value}
            if not args.no_cpu_baseline:
                import oracle
                oracle.set_num_threads(host_cpu_info()["usable"])
                chk = {}
                for name_, sets_, ratio_ in (("hard_ratio_0.5", hsets, 0.5), ("hard_ratio_0.8", hsets, 0.8), ("m_surf_4k_ratio_0.8", sets, 0.8)):
                    ref_ = oracle.match_pairs_l2(sets_, hpairs, ratio_)
                    chk[name_] = all(np.array_equal(a_[0], b_[0]) and np.array_equal(a_[1], b_[1]) and np.array_equal(a_[2].view(np.uint32), b_[2].view(np.uint32))
                                     for a_, b_ in zip(res_h[name_], ref_))
                out["hard"]["verified_vs_oracle"] = all(chk.values())
                out["hard"]["verified_scope"] = "all 300 pairs of each of the three legs: every (queryIdx, trainIdx, distance bits), match lists in order"
                out["hard"]["verified_legs"] = chk
            roofline["legs"]["hard"] = {"value": out["hard"]["value"], "unit": "image-pairs/s (M-SURF-4k-hard, ratio 0.5)", "vs_benign": out["hard"]["vs_benign"],
                                        "finish_kernel_ms": legs_h["hard_ratio_0.5"]["finish_kernel_ms"],
                                        "screen_survivor_frac": legs_h["hard_ratio_0.5"]["screen_survivors_per_step"] / max(legs_h["hard_ratio_0.5"]["queries_per_step"], 1),
                                        "second_pass_frac": legs_h["hard_ratio_0.5"]["second_pass_queries_per_step"] / max(legs_h["hard_ratio_0.5"]["queries_per_step"], 1),
                                        "ratio_0.8_value": legs_h["hard_ratio_0.8"]["value"], "m_surf_4k_ratio_0.8_value": legs_h["m_surf_4k_ratio_0.8"]["value"],
                                        "verified_vs_oracle": out["hard"].get("verified_vs_oracle")}
        except Exception as e:
            out["hard"] = {"error": repr(e)}

    # ---------------------------------------------------------------- sparse-cloud outlier filter (row f-3; one cloud: rank 0 at N = 1)
    if rank == 0 and world == 1 and not args.no_ba:
        try:
            import ctypes as C
            rng = np.random.default_rng(4100)
            cloud = np.concatenate([synth.ba_scene(25, 30000, 8, radius=10.0, extent=2.0, seed=4000).pts_gt,
                                    rng.uniform(-12, 12, (600, 3))]).astype(np.float32)           # BA-25 cloud + 2 % far points
            n_c = len(cloud)
            d_pts = torch.from_numpy(cloud).to(dev); d_out = torch.empty(n_c, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            sctx = E.Context(local_rank, None)
            L = E.lib()
            call = lambda: _lib.check(L.esfm_sor_mean_distances_dev(sctx.handle, C.c_void_p(d_pts.data_ptr()), n_c, 3, 50,
                                                                    C.c_void_p(d_out.data_ptr())))
            call(); sctx.synchronize()
            sctx.set_kernel_timing(True); sctx.kernel_time(_lib.K_SOR_KNN)
            for _ in range(10):
                call()
            sctx.synchronize()
            k_ms2, k_n2 = sctx.kernel_time(_lib.K_SOR_KNN)
            sctx.set_kernel_timing(False)
            t_s = k_ms2 / max(k_n2, 1) * 1e-3
            # 3 sub + 3 mul + 2 add + 1 compare per candidate (SURVEY 8d-style VALU count) for the ALGORITHMIC N^2 candidates: from 4096
            # points on the kernel sweeps a window of the sorted cloud and evaluates a fraction of them (since round 3), so this is
            # brute-force-equivalent work per second, not what the VALU executed
            lane_ops = 9.0 * n_c * n_c
            keep, md, thr = E.sor_filter(cloud, 50, 2.0, sctx)
            out["cloud"] = {"metric": "SOR filter points/s (MeanK 50)", "value": n_c / t_s, "unit": "points/s", "points": n_c,
                            "kept": int(keep.sum()), "kernel": "sort + sor_knn_mean_kernel<sorted window>", "avg_launch_ms": t_s * 1e3,
                            "brute_force_equivalent_pair_evaluations_per_s": n_c * float(n_c) / t_s,
                            "roofline": {"bound": "valu", "achieved": lane_ops / t_s / 1e12, "peak": 78.6, "unit": "T lane-ops/s (brute-force-equivalent)",
                                         "frac": lane_ops / t_s / 1e12 / 78.6}}
            if not args.no_cpu_baseline:
                import oracle
                oracle.sor_filter(cloud[:2048], 50, 2.0)      # thread pool warm-up
                t0 = time.perf_counter(); rk, rmd, rthr = oracle.sor_filter(cloud, 50, 2.0); t1 = time.perf_counter() - t0
                out["cloud"]["verified_vs_oracle"] = bool(np.array_equal(md, rmd) and np.array_equal(keep, rk) and thr == rthr)
                out["cloud"]["cpu_baseline"] = {"value": n_c / t1, "unit": "points/s", "cores": oracle.num_threads(), "kind": "port",
                                                "sample": f"the same {n_c}-point cloud in {t1:.2f}s (brute-force k-NN, OpenMP over points)"}
        except Exception as e:
            out["cloud"] = {"error": repr(e)}

    # ---------------------------------------------------------------- SURF from pixels (row f-2), rank 0 at N = 1
    if rank == 0 and world == 1 and not args.no_ba:
        try:
            gold = os.path.join(ROOT, "tests", "golden", "fountain11_half_gray.npz")
            half = np.load(gold)["images"][0]
            img = np.ascontiguousarray(np.kron(half, np.ones((2, 2), np.uint8)))               # 512 x 768 like the reference's images
            img = np.clip(img.astype(np.int16) + np.random.default_rng(4300).integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
            fctx = E.Context(local_rank, None)
            kp_, d_ = E.surf_detect_and_compute(img, 300.0, None, fctx)
            fctx.set_kernel_timing(True); fctx.kernel_time(_lib.K_SURF_DET); fctx.kernel_time(_lib.K_SURF_DESC)
            t0 = time.perf_counter()
            n_rep = 10
            for _ in range(n_rep):
                kp_, d_ = E.surf_detect_and_compute(img, 300.0, None, fctx)
            f_el = (time.perf_counter() - t0) / n_rep
            dt_ms, dt_n = fctx.kernel_time(_lib.K_SURF_DET); ds_ms, ds_n = fctx.kernel_time(_lib.K_SURF_DESC)
            fctx.set_kernel_timing(False)
            out["surf"] = {"metric": "images/s, SURF detect + describe (768 x 512, minHessian 300)", "value": 1.0 / f_el, "unit": "images/s",
                           "keypoints": int(len(kp_)), "ms_per_image": f_el * 1e3, "det_trace_kernel_ms": dt_ms / max(dt_n, 1),
                           "describe_kernel_ms": ds_ms / max(ds_n, 1), "includes": "host<->device copies, host sort of the maxima"}
            if not args.no_cpu_baseline:
                import oracle
                t0 = time.perf_counter(); rk_, rd_ = oracle.surf(img, 300.0); t1 = time.perf_counter() - t0
                out["surf"]["verified_vs_oracle"] = bool(np.array_equal(kp_.view(np.uint32), rk_.view(np.uint32)) and
                                                         np.array_equal(d_.view(np.uint32), rd_.view(np.uint32)))
                out["surf"]["cpu_baseline"] = {"value": 1.0 / t1, "unit": "images/s", "cores": 1, "kind": "port",
                                               "sample": f"the same image in {t1:.2f}s (sequential restatement, one core)"}
        except Exception as e:
            out["surf"] = {"error": repr(e)}

    # ---------------------------------------------------------------- image undistortion (row f-2), rank 0 at N = 1
    if rank == 0 and world == 1 and not args.no_ba:
        try:
            u_rows, u_cols = 2048, 3072                                     # the reference's image size (fountain-P11)
            uimg = np.random.default_rng(4400).integers(0, 256, (u_rows, u_cols, 3), dtype=np.uint8)
            uK, udist = [2759.48, 1520.69, 2764.16, 1006.81], [-0.12, 0.03, 0.001, -0.0005]
            uctx = E.Context(local_rank, None)
            uo = E.undistort(uimg, uK, udist, uctx)
            uctx.set_kernel_timing(True); uctx.kernel_time(_lib.K_UNDISTORT)
            t0 = time.perf_counter()
            n_rep = 10
            for _ in range(n_rep):
                uo = E.undistort(uimg, uK, udist, uctx)
            u_el = (time.perf_counter() - t0) / n_rep
            u_ms, u_n = uctx.kernel_time(_lib.K_UNDISTORT)
            uctx.set_kernel_timing(False)
            u_s = u_ms / max(u_n, 1) * 1e-3
            u_bytes = 6.0 * u_rows * u_cols                                 # 3 B read + 3 B written per pixel
            # the same call on registered (pinned) host buffers: one DMA transfer each way instead of 512-KiB pieces through the
            # runtime's staging buffer (easysfm_amd/csrc/common.hpp copy_h2d: why pageable buffers are not handed over whole)
            u_pin = None
            try:
                import ctypes as C_
                pin_i = torch.empty((u_rows, u_cols, 3), dtype=torch.uint8).pin_memory(); pin_i.numpy()[:] = uimg
                pin_o = torch.empty((u_rows, u_cols, 3), dtype=torch.uint8).pin_memory()
                k4_ = np.array(uK, np.float64); d4_ = np.array(udist, np.float64)

                def call_pinned():
                    rc_ = _lib.lib().esfm_undistort(uctx.handle, C_.c_void_p(pin_i.data_ptr()), u_rows, u_cols, 3, C_.c_void_p(k4_.ctypes.data),
                                                    C_.c_void_p(d4_.ctypes.data), C_.c_void_p(pin_o.data_ptr()))
                    if rc_ != 0:
                        raise RuntimeError("esfm_undistort on pinned buffers failed")
                call_pinned()
                t0 = time.perf_counter()
                for _ in range(n_rep):
                    call_pinned()
                u_pin = (time.perf_counter() - t0) / n_rep
                if not np.array_equal(pin_o.numpy(), uo):
                    u_pin = None
            except Exception:
                u_pin = None
            out["undistort"] = {"metric": "images/s, cv::undistort 3072 x 2048 BGR (k1 k2 p1 p2)", "value": 1.0 / u_el, "unit": "images/s",
                                "ms_per_image": u_el * 1e3,
                                "includes": "host<->device copies of the image from / to pageable host memory (in 512-KiB pieces: host-copy-bound)",
                                "ms_per_image_registered_host_memory": (u_pin * 1e3) if u_pin else None,
                                "kernel": "undistort_remap_kernel", "avg_launch_ms": u_s * 1e3,
                                "roofline": {"bound": "hbm", "achieved": u_bytes / u_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                                             "frac": u_bytes / u_s / 1e9 / 8000.0, "traffic": None}}
            if not args.no_cpu_baseline:
                import oracle
                t0 = time.perf_counter(); ur = oracle.undistort(uimg, uK, udist); t1 = time.perf_counter() - t0
                out["undistort"]["verified_vs_oracle"] = bool(np.array_equal(uo, ur))
                out["undistort"]["cpu_baseline"] = {"value": 1.0 / t1, "unit": "images/s", "cores": 1, "kind": "port",
                                                    "sample": f"the same image in {t1:.2f}s (sequential restatement, one core)"}
        except Exception as e:
            out["undistort"] = {"error": repr(e)}

    # ---------------------------------------------------------------- geometric verification (row f-1), rank 0 at N = 1
    if rank == 0 and world == 1 and not args.no_ba:
        try:
            rng = np.random.default_rng(4200)
            K4v = np.array(synth.FOUNTAIN_K4, np.float32)

            def two_view(n, frac):
                R = synth.aa_to_R(rng.normal(0, 0.15, 3)); t = np.array([1.0, 0.1, -0.05]) + rng.normal(0, 0.05, 3); t /= np.linalg.norm(t)
                X = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0, 8.0]); Xc = X @ R.T + t
                a = (X[:, :2] / X[:, 2:3] * [K4v[0], K4v[2]] + [K4v[1], K4v[3]]).astype(np.float32)
                b = (Xc[:, :2] / Xc[:, 2:3] * [K4v[0], K4v[2]] + [K4v[1], K4v[3]] + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
                bad = rng.choice(n, int(frac * n), replace=False); b[bad] += rng.uniform(-60, 60, (len(bad), 2)).astype(np.float32)
                return a, b

            n_pairs_g, n_m = 300, 1000
            jobs = [two_view(n_m, 0.3) for _ in range(n_pairs_g)]
            off = np.arange(n_pairs_g + 1, dtype=np.int32) * n_m
            pa = np.concatenate([j[0] for j in jobs]); pb_ = np.concatenate([j[1] for j in jobs]); Ks = np.tile(K4v, (n_pairs_g, 1))
            gctx = E.Context(local_rank, None)
            E.find_essential_pairs(off, pa, pb_, Ks, 0.99, 1.0, gctx)
            t0 = time.perf_counter()
            Es_, mask_, st_, it_ = E.find_essential_pairs(off, pa, pb_, Ks, 0.99, 1.0, gctx)
            good_, Rs_, ts_, _m = E.recover_pose_pairs(off, pa, pb_, Ks, Es_, mask_, gctx)
            g_el = time.perf_counter() - t0
            out["geometry"] = {"metric": "image pairs verified/s (5-point RANSAC + recoverPose, 1000 matches, 30% outliers)",
                               "value": n_pairs_g / g_el, "unit": "image-pairs/s", "pairs": n_pairs_g, "matches_per_pair": n_m,
                               "mean_ransac_iterations": float(it_.mean()), "batch_ms": g_el * 1e3, "includes": "host<->device copies of the batch"}
            if not args.no_cpu_baseline:
                import oracle
                t0 = time.perf_counter()
                same = True
                for k in range(24):
                    ok_, Er_, mr_, itr_, cnt_ = oracle.find_essential_ransac(jobs[k][0], jobs[k][1], K4v, 0.99, 1.0)
                    oracle.recover_pose(Er_, jobs[k][0], jobs[k][1], K4v, mr_)
                    same = same and itr_ == int(it_[k]) and np.array_equal(mr_, mask_[off[k]:off[k + 1]]) and np.array_equal(Er_, Es_[k])
                t1 = time.perf_counter() - t0
                out["geometry"]["verified_vs_oracle"] = bool(same)
                out["geometry"]["cpu_baseline"] = {"value": 24 / t1, "unit": "image-pairs/s", "cores": 1, "kind": "port",
                                                   "sample": f"24 of the pairs in {t1:.2f}s (sequential RANSAC restatement, one core)"}
        except Exception as e:
            out["geometry"] = {"error": repr(e)}

    # ---------------------------------------------------------------- the "next" rows that had no leg yet (f-1: PnP RANSAC, triangulation; f-2: ORB), rank 0 at N = 1
    if rank == 0 and world == 1 and not args.no_ba:
        try:
            gold = os.path.join(ROOT, "tests", "golden", "fountain11_gray.npz")
            oimg = np.load(gold)["images"][3]
            xctx = E.Context(local_rank, None)
            okp, odesc = E.orb_detect_and_compute(oimg, 8000, None, xctx)
            t0 = time.perf_counter()
            for _ in range(10):
                okp, odesc = E.orb_detect_and_compute(oimg, 8000, None, xctx)
            o_el = (time.perf_counter() - t0) / 10
            out["orb_detect"] = {"metric": "images/s, ORB detect + describe (768 x 512, 8000 features asked for)", "value": 1.0 / o_el, "unit": "images/s",
                                 "keypoints": int(len(okp)), "ms_per_image": o_el * 1e3, "includes": "host<->device copies, host selection of the best responses"}
            if not args.no_cpu_baseline:
                import oracle
                t0 = time.perf_counter(); rk_, rd_ = oracle.orb(oimg, 8000); t1 = time.perf_counter() - t0
                out["orb_detect"]["verified_vs_oracle"] = bool(np.array_equal(okp.view(np.uint32), rk_.view(np.uint32)) and np.array_equal(odesc, rd_))
                out["orb_detect"]["cpu_baseline"] = {"value": 1.0 / t1, "unit": "images/s", "cores": 1, "kind": "port", "sample": f"the same image in {t1:.2f}s (sequential restatement, one core)"}
        except Exception as e:
            out["orb_detect"] = {"error": repr(e)}
        try:
            rng = np.random.default_rng(4500)
            K4p = np.array(synth.FOUNTAIN_K4, np.float32)
            n3 = 2000
            Rt = synth.aa_to_R(rng.normal(0, 0.2, 3)); tt_ = np.array([0.3, -0.1, 0.5])
            X3 = (rng.uniform(-2, 2, (n3, 3)) + np.array([0, 0, 8.0])).astype(np.float32)
            Xc = X3.astype(np.float64) @ Rt.T + tt_
            pix = np.stack([Xc[:, 0] / Xc[:, 2] * K4p[0] + K4p[1], Xc[:, 1] / Xc[:, 2] * K4p[2] + K4p[3]], 1) + rng.normal(0, 0.5, (n3, 2))
            bad = rng.choice(n3, n3 // 4, replace=False); pix[bad] += rng.uniform(-80, 80, (len(bad), 2))
            pix = pix.astype(np.float32)
            pctx = E.Context(local_rank, None)
            E.solve_pnp_ransac(X3, pix, K4p, 100, 8.0, 0.99, pctx)
            t0 = time.perf_counter()
            for _ in range(10):
                rv_, tv_, R_, m_, it_p = E.solve_pnp_ransac(X3, pix, K4p, 100, 8.0, 0.99, pctx)
            p_el = (time.perf_counter() - t0) / 10
            out["pnp"] = {"metric": "frames registered/s (solvePnPRansac EPnP, 2000 2-D/3-D pairs, 25 % outliers)", "value": 1.0 / p_el, "unit": "frames/s",
                          "ms_per_frame": p_el * 1e3, "ransac_iterations": int(it_p), "inliers": int(m_.sum()), "includes": "host<->device copies, host replay of OpenCV's RNG"}
            if not args.no_cpu_baseline:
                import oracle
                t0 = time.perf_counter(); ok_, rR, rt_, rrv, rm, rit = oracle.solve_pnp_ransac(X3, pix, K4p, 100, 8.0, 0.99); t1 = time.perf_counter() - t0
                out["pnp"]["verified_vs_oracle"] = bool(ok_ and rit == it_p and np.array_equal(rm, m_) and np.allclose(rR, R_, atol=1e-9) and np.allclose(rt_, tv_, atol=1e-9))
                out["pnp"]["cpu_baseline"] = {"value": 1.0 / t1, "unit": "frames/s", "cores": 1, "kind": "port", "sample": f"the same problem in {t1 * 1e3:.1f} ms (sequential restatement, one core)"}
            # two-view triangulation of the inliers (cv::triangulatePoints: DLT per point)
            P1 = np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float32); P2 = np.hstack([Rt, tt_[:, None]]).astype(np.float32)
            n_t = 200000
            Xt = rng.uniform(-2, 2, (n_t, 3)) + np.array([0, 0, 8.0]); Xt2 = Xt @ Rt.T + tt_
            a_ = (Xt[:, :2] / Xt[:, 2:3]).astype(np.float32); b_ = (Xt2[:, :2] / Xt2[:, 2:3]).astype(np.float32)
            E.triangulate_points(P1, P2, a_, b_, pctx)
            t0 = time.perf_counter()
            for _ in range(10):
                X4 = E.triangulate_points(P1, P2, a_, b_, pctx)
            t_el = (time.perf_counter() - t0) / 10
            out["triangulate"] = {"metric": "points triangulated/s (cv::triangulatePoints, two views)", "value": n_t / t_el, "unit": "points/s", "points": n_t,
                                  "ms_per_call": t_el * 1e3, "includes": "host<->device copies from pageable memory in 512-KiB pieces (host-copy-bound: 16 B in, 16 B out per point)"}
            if not args.no_cpu_baseline:
                import oracle
                t0 = time.perf_counter(); rX4 = oracle.triangulate_points(P1, P2, a_, b_); t1 = time.perf_counter() - t0
                out["triangulate"]["verified_vs_oracle"] = bool(np.array_equal(X4.view(np.uint32), rX4.view(np.uint32)))       # (one Jacobi rule on both sides: bit for bit)
                out["triangulate"]["cpu_baseline"] = {"value": n_t / t1, "unit": "points/s", "cores": 1, "kind": "port", "sample": f"the same {n_t} points in {t1:.2f}s (sequential restatement, one core)"}
        except Exception as e:
            out.setdefault("pnp", {"error": repr(e)})
            out.setdefault("triangulate", {"error": repr(e)})

    # ---------------------------------------------------------------- BASELINE configs 1 and 3 end to end (rank 0 at N = 1)
    if rank == 0 and world == 1 and not args.no_ba and not args.no_e2e:
        for tag, feat, par in (("config1", "S", 300), ("config3", "O", 8000)):
            try:
                out[tag] = e2e_leg(tag, feat, par, not args.no_cpu_baseline)
            except Exception as e:
                out[tag] = {"error": repr(e)}

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"], ver = cpu_baseline_match(sets, pairs, res_host)
            if ver is not None:
                out["verified_vs_oracle"] = bool(ver["ok"])
                out["verified_scope"] = (f"all {ver['pairs_checked']} of {ver['pairs_per_step']} pairs of the step "
                                         f"({ver['queries_checked']} queries): {ver['what']}")
                if ver["mismatching_pairs"]:
                    out["verified_mismatching_pairs"] = ver["mismatching_pairs"]
                roofline["parity"]["verified_vs_oracle"] = out["verified_vs_oracle"]
                roofline["parity"]["verified_scope"] = out["verified_scope"]
            if "ba" in out and "error" not in out["ba"]:
                out["ba"]["cpu_baseline"] = cpu_baseline_ba(synth.ba_scene(25, 30000, 8, radius=10.0, extent=2.0, seed=4000))
                out["cpu_baseline"]["ba"] = out["ba"]["cpu_baseline"]      # the BA half beside the matching half, where the driver's record keeps it
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    emit()
    if isinstance(comm, E.Comm):
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

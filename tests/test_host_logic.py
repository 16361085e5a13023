"""Host-side logic above the C ABI that needs no GPU: pair-list / point sharding and the reference's
observation derivation (setBAProblem, ba.cpp:22-56).  CPU only."""
import os
import sys

import numpy as np
import pytest

import easysfm_amd as E
from easysfm_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pair_list_is_the_reference_loop():
    p = E.shard_pair_list(6, None, 0, 1)
    assert p.tolist() == [[i, j] for i in range(6) for j in range(i)]     # sfm.cpp:140-143: query i, train j < i
    assert np.array_equal(p, synth.all_pairs(6))
    assert len(E.shard_pair_list(1, None, 0, 1)) == 0 and len(E.shard_pair_list(0, None, 0, 1)) == 0


@pytest.mark.parametrize("world", [2, 3, 8])
def test_pair_shards_partition_and_balance(world):
    rows = np.random.default_rng(world).integers(500, 5000, 40).astype(np.int32)
    shards = [E.shard_pair_list(40, rows, r, world) for r in range(world)]
    allp = np.concatenate(shards)
    assert len(allp) == 40 * 39 // 2
    assert len({tuple(x) for x in allp.tolist()}) == len(allp)            # disjoint, complete
    cost = [float((rows[s[:, 0]].astype(np.float64) * rows[s[:, 1]]).sum()) for s in shards]
    assert max(cost) - min(cost) <= float(rows.max()) ** 2                 # within one pair's cost
    for s in shards:                                                       # reference order kept inside a shard
        key = s[:, 0].astype(np.int64) * 1000 + s[:, 1]
        assert np.all(np.diff(key) > 0)


def test_point_shards_balance_observations():
    sc = synth.ba_scene(10, 1000, 6, seed=1)
    for world in (2, 4, 8):
        sh = E.shard_points(sc.n_pt, sc.pt_idx, world)
        assert sh.min() == 0 and sh.max() == world - 1 and np.all(np.diff(sh) >= 0)
        cnt = np.bincount(sh[sc.pt_idx], minlength=world)
        assert cnt.sum() == sc.n_obs and cnt.max() - cnt.min() <= 2 * 6


def _triple_loop(frames, process, cloud):
    """The reference's own O(Ncam*Npts*Nkp) derivation, ba.cpp:22-56, verbatim semantics."""
    cams, pts, uv = [], [], []
    ncam = 0
    for i, fr in enumerate(frames):
        if process[i]:
            continue
        for k in range(len(cloud.unique_point_ids)):
            for j in range(len(fr.unique_pixel_ids)):
                if fr.unique_pixel_has_match[j] and fr.unique_pixel_ids[j] == cloud.unique_point_ids[k]:
                    uv.append(fr.keypoints[j]); pts.append(k); cams.append(ncam)
                    break
        ncam += 1
    return np.array(cams, np.int32), np.array(pts, np.int32), np.array(uv, np.float32).reshape(-1, 2)


def test_setBAProblem_matches_reference_triple_loop():
    rng = np.random.default_rng(4)
    frames, process = [], []
    for i in range(5):
        n = int(rng.integers(20, 60))
        fr = E.Frame(frame_id=i, keypoints=rng.uniform(0, 700, (n, 2)).astype(np.float32))
        fr.unique_pixel_ids = rng.integers(0, 40, n)          # many duplicates: the `break` picks the first
        fr.unique_pixel_has_match = rng.random(n) < 0.6
        fr.pose_cam = np.eye(4, dtype=np.float32); fr.pose_cam[:3, 3] = rng.standard_normal(3)
        fr.K_cam = np.array([[700, 0, 380], [0, 701, 250], [0, 0, 1]], np.float32)
        frames.append(fr); process.append(bool(i == 3))
    cloud = E.SparsePointCloud(xyz=rng.standard_normal((35, 3)).astype(np.float32),
                               unique_point_ids=rng.permutation(45)[:35])
    ba = E.BundleAdjustment.__new__(E.BundleAdjustment)
    ba._ctx = None; ba.options = None
    ba.initBA()
    ready = ba.setBAProblem(frames, process, cloud)
    c, p, uv = _triple_loop(frames, process, cloud)
    assert ba.num_cameras_ == 4 and ba.num_points_ == 35
    assert np.array_equal(ba.camera_index_, c) and np.array_equal(ba.point_index_, p) and np.array_equal(ba.points_2d_, uv)
    assert ba.num_parameters_ == 6 * 4 + 3 * 35 and ready == (2 * len(c) > ba.num_parameters_)
    # parameter packing (ba.cpp:70-104): angle-axis of the float pose, translation, then float points
    assert np.allclose(ba.parameters_[3:6], frames[0].pose_cam[:3, 3]) and np.allclose(ba.parameters_[:3], 0)
    assert np.array_equal(ba.parameters_[24:].reshape(-1, 3), cloud.xyz.astype(np.float64))


def test_rodrigues_roundtrip():
    from easysfm_amd.ba import angle_axis_to_rotation, rotation_to_angle_axis
    rng = np.random.default_rng(0)
    for _ in range(50):
        aa = rng.standard_normal(3) * rng.uniform(0, 3.0)
        R = angle_axis_to_rotation(aa)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
        assert np.allclose(angle_axis_to_rotation(rotation_to_angle_axis(R)), R, atol=1e-9)
    assert np.allclose(rotation_to_angle_axis(np.eye(3)), 0)


def test_free_calib_needs_the_gpu_library():
    """No CPU fallback: without a GPU the free-intrinsics solve raises from the C ABI instead of computing elsewhere."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ba = E.BundleAdjustment.__new__(E.BundleAdjustment)
    ba._ctx = None; ba.options = None
    ba.initBA()
    ba.parameters_ = np.array([700.0, 380.0, 700.0, 250.0])
    with pytest.raises(E.EsfmError):
        ba.solveBA(20.0)


def test_line_search_step_rule_matches_oracle(oracle_lib):
    """The Armijo step-length rule (cubic / quintic interpolation, polynomial.cc) is host arithmetic on both sides:
    product (esfm_ba_line_search_next_step) == oracle on random samples, including invalid trials."""
    from easysfm_amd.ba import line_search_next_step
    rng = np.random.default_rng(5)
    n_poly = 0
    for it in range(400):
        f0 = rng.uniform(1, 100); g0 = -rng.uniform(0.1, 50)
        xc = rng.uniform(1e-3, 1.0)
        cur = (xc, f0 + rng.uniform(-1, 5) * abs(g0) * xc, rng.uniform(-3, 3) * abs(g0))
        prev = None
        if it % 3 == 0:
            xp = xc / rng.uniform(0.05, 0.6)
            prev = (xp, f0 + rng.uniform(-1, 5) * abs(g0) * xp, rng.uniform(-3, 3) * abs(g0))
        if it % 17 == 0:
            # an invalid trial carries only its step size: both halve it
            got = E._lib.lib().esfm_ba_line_search_next_step(f0, g0, 0.0, 0.0, 0.0, 0, xc, 0.0, 0.0, 0)
            ref = oracle_lib.load().esfm_ref_ls_next_step(f0, g0, 0.0, 0.0, 0.0, 0, xc, 0.0, 0.0, 0, 1e-3 * xc, 0.6 * xc)
            assert got == pytest.approx(0.5 * xc) and ref == pytest.approx(0.5 * xc)
            continue
        got = line_search_next_step(f0, g0, prev, cur)
        ref = oracle_lib.ls_next_step(f0, g0, prev, cur, 1e-3 * xc, 0.6 * xc)
        assert 1e-3 * xc <= got <= 0.6 * xc
        assert got == pytest.approx(ref, rel=1e-9, abs=1e-12), (it, f0, g0, prev, cur)
        n_poly += 1
    assert n_poly > 300


def _init_pair_loops(T, depth, min_track, max_ratio):
    """feature_matching.cpp:160-229, the reference's loops verbatim."""
    n_frames, n_unique = T.shape
    cnt = [0] * n_unique
    for i in range(n_frames):
        for j in range(n_unique):
            cnt[j] += int(T[i][j])
    best, f1, f2 = min_track, 0, 0
    for i in range(n_frames):
        for j in range(i):
            if depth[i][j] > max_ratio:
                continue
            s = 0
            for k in range(n_unique):
                if T[i][k] and T[j][k]:
                    s += cnt[k]
            if s >= best:
                best, f1, f2 = s, i, j
    return f1, f2


def test_frame_selection_matches_reference_loops():
    rng = np.random.default_rng(3)
    fm = E.FeatureMatching.__new__(E.FeatureMatching)
    for trial in range(20):
        F, U = int(rng.integers(3, 9)), int(rng.integers(5, 60))
        T = rng.random((F, U)) < 0.4
        depth = rng.uniform(1, 150, (F, F))
        frames = [None] * F
        found, f1, f2, d = fm.findInitializeFramePair(T, frames, depth, min_track_num_init=5, max_depth_baseline_ratio_init=100.0)
        r1, r2 = _init_pair_loops(T, depth, 5, 100.0)
        if r1 == r2:
            assert not found and (f1, f2) == (1, 0)
        else:
            assert found and (f1, f2) == (r1, r2) and d == depth[r1][r2]
        todo = list(rng.random(F) < 0.5)
        ids = rng.permutation(U)[:int(rng.integers(0, U))]
        nxt = fm.findNextFrame(T, todo, ids, next_frame=-7)
        best, ref = 0, -7
        for i in range(F):
            if todo[i]:
                c = sum(1 for k in ids if T[i][k])
                if c > best:
                    best, ref = c, i
        assert nxt == ref


def test_ply_writer_layout(tmp_path):
    """The external contract argv[5] (data_io.cpp:147-165): PCL's ASCII PLY of PointXYZRGB, camera element included."""
    cloud = E.SparsePointCloud(xyz=np.array([[1.5, -2.25, 3.0], [0.1, 0.2, 0.3]], np.float32), rgb=np.array([[255, 0, 7], [1, 2, 3]], np.uint8))
    path = str(tmp_path / "out.ply")
    assert E.write_ply(path, cloud)
    text = open(path).read().split("\n")
    assert text[:4] == ["ply", "format ascii 1.0", "comment PCL generated", "element vertex 2"]
    assert text[4:10] == ["property float x", "property float y", "property float z", "property uchar red", "property uchar green", "property uchar blue"]
    assert text[10] == "element camera 1" and text[text.index("end_header") + 1] == "1.5 -2.25 3 255 0 7"
    assert text[text.index("end_header") + 2] == "0.1 0.2 0.30000001 1 2 3"      # float printed with 8 significant digits
    xyz, rgb, cam = E.read_ply_vertices(path)
    assert np.array_equal(xyz, cloud.xyz) and np.array_equal(rgb, cloud.rgb)
    assert cam == "0 0 0 1 0 0 0 1 0 0 0 1 0 0 0 0 0 1 2 0 0"                    # viewport = width 1 x height N (:151-152)


def test_ransac_sample_stream_matches_oracle(oracle_lib):
    """cv::RNG((uint64)-1) + getSubset replayed on the host on both sides: identical 5-index samples."""
    for count in (5, 6, 37, 1000, 4096):
        a = E.ransac_sample_stream(count, 300)
        b = oracle_lib.ransac_samples(count, 300)
        assert np.array_equal(a, b)
        assert a.min() >= 0 and a.max() < count and all(len(set(r)) == 5 for r in a.tolist())
    # the recurrence itself: state' = lo32(state) * 4164903690 + hi32(state)
    s = 0xFFFFFFFFFFFFFFFF
    s = (s & 0xFFFFFFFF) * 4164903690 + (s >> 32)
    assert E.ransac_sample_stream(1000, 1)[0, 0] == (s & 0xFFFFFFFF) % 1000


def test_bin_sfm_command_line(tmp_path):
    """./bin/sfm keeps the reference's 13 positional arguments (sfm.cpp:35-50); feature type O (ORB) is accepted like S; a missing
    input fails with a non-success status (1 is the reference's SUCCESS status, sfm.cpp:339)."""
    import os
    import subprocess
    import sys
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "sfm")
    r = subprocess.run([sys.executable, exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 2 and "feature_type" in r.stdout
    args = ["imgs", "list.txt", "K.txt", "none", str(tmp_path / "o.ply"), "O", "8000", "1.0", "1", "0", "4", "0", "0"]
    r = subprocess.run([sys.executable, exe] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode not in (0, 1) and "list.txt" in r.stdout and "ORB" not in r.stdout
    sys.path.insert(0, os.path.dirname(exe))
    import importlib.machinery
    import importlib.util
    loader = importlib.machinery.SourceFileLoader("bin_sfm", exe)
    mod = importlib.util.module_from_spec(importlib.util.spec_from_loader("bin_sfm", loader))
    loader.exec_module(mod)
    (tmp_path / "list.txt").write_text("0000.png\r\n0001.png\n\n")
    (tmp_path / "K.txt").write_text("689.87 0 380.17\r\n0 691.04 251.70\r\n0 0 1")
    assert mod.import_image_filenames(str(tmp_path / "list.txt"), "dir") == [os.path.join("dir", "0000.png"), os.path.join("dir", "0001.png")]
    K = mod.import_calib(str(tmp_path / "K.txt"))
    assert K.shape == (3, 3) and abs(K[0, 2] - 380.17) < 1e-4 and K[2, 2] == 1


def test_import_distort_reproduces_the_float_into_double_storage(tmp_path):
    """DataIO::importDistort (data_io.cpp:97-125) writes floats into a CV_64F matrix: the doubles OpenCV
    sees are the float pairs reinterpreted; a missing file leaves the default zeros to the caller."""
    import struct
    import easysfm_amd as E
    p = tmp_path / "dist.txt"
    p.write_text("-0.28 1.35 0.0007 -0.0004\n")
    c = E.import_distort(str(p))
    assert c.dtype == np.float64 and c.shape == (4,)
    k1p = struct.unpack("<d", struct.pack("<ff", np.float32(-0.28), np.float32(1.35)))[0]
    k2p = struct.unpack("<d", struct.pack("<ff", np.float32(0.0007), np.float32(-0.0004)))[0]
    assert c[0] == k1p and c[1] == k2p and c[2] == 0 and c[3] == 0
    assert abs(c[0] - 0.05625) < 1e-6                      # k2 = 1.35 lands in the exponent: a visible k1'
    p.write_text("1 2 3 4\n5 6")                          # a second, partial group overwrites k1 k2 only
    c = E.import_distort(str(p))
    assert np.array_equal(c.view(np.float32)[:4], np.array([5, 6, 3, 4], np.float32))
    assert E.import_distort(str(tmp_path / "none")) is None


def _build_native(tmp_path):
    import subprocess
    exe = str(tmp_path / "sfm_native")
    cmd = ["g++", "-O2", "-std=c++17", os.path.join(ROOT, "easysfm_amd", "host", "sfm_main.cpp"), "-o", exe,
           os.path.join(ROOT, "easysfm_amd", "libesfm_hip.so"), "-lz", "-Wl,-rpath," + os.path.join(ROOT, "easysfm_amd"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_native_png_reader_matches_pil(tmp_path):
    """The C++ executable's PNG reader (easysfm_amd/host/esfm_png.hpp, cv::imread(..., COLOR) for the reference's inputs): RGB,
    gray, RGBA, palette and 16-bit files come out as the same 8-bit BGR pixels PIL decodes; a non-PNG is refused."""
    import subprocess
    PIL = pytest.importorskip("PIL.Image")
    exe = _build_native(tmp_path)
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    rgb[5:20, 10:40] = (rgb[5:20, 10:40] // 32) * 32                 # smooth areas: exercises the sub / up / average / Paeth filters
    cases = {
        "rgb.png": (PIL.fromarray(rgb), rgb[:, :, ::-1]),
        "gray.png": (PIL.fromarray(rgb[:, :, 0]), np.stack([rgb[:, :, 0]] * 3, axis=2)),
        "rgba.png": (PIL.fromarray(np.dstack([rgb, rng.integers(0, 256, (37, 53), dtype=np.uint8)]), "RGBA"), rgb[:, :, ::-1]),
    }
    pal = PIL.fromarray(rgb).quantize(16)
    cases["palette.png"] = (pal, np.asarray(pal.convert("RGB"))[:, :, ::-1])
    bw = PIL.fromarray(rgb[:, :, 1]).convert("1")
    cases["bilevel.png"] = (bw, np.stack([np.asarray(bw.convert("L"))] * 3, axis=2))
    g16 = (rng.integers(0, 65536, (21, 34))).astype(np.uint16)
    cases["gray16.png"] = (PIL.fromarray(g16), np.stack([(g16 >> 8).astype(np.uint8)] * 3, axis=2))
    for name, (im, want) in cases.items():
        path = str(tmp_path / name)
        im.save(path)
        out = str(tmp_path / (name + ".raw"))
        r = subprocess.run([exe, "--dump-image", path, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, (name, r.stdout)
        raw = np.fromfile(out, np.uint8)
        rows, cols = np.frombuffer(raw[:8].tobytes(), np.int32)
        got = raw[8:].reshape(rows, cols, 3)
        assert np.array_equal(got, np.ascontiguousarray(want)), name
    good = (tmp_path / "rgb.png").read_bytes()
    for name, data, msg in (("trunc.png", good[:len(good) // 2], "truncated"), ("crc.png", good[:60] + bytes([good[60] ^ 0xFF]) + good[61:], "CRC")):
        (tmp_path / name).write_bytes(data)
        r = subprocess.run([exe, "--dump-image", str(tmp_path / name), str(tmp_path / "x.raw")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 3 and (msg in r.stdout or "inflate" in r.stdout or "header" in r.stdout), (name, r.stdout)
    bad = tmp_path / "not.png"
    bad.write_bytes(b"JFIF" * 10)
    r = subprocess.run([exe, "--dump-image", str(bad), str(tmp_path / "x.raw")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 3 and "not a PNG" in r.stdout


def test_native_png_reader_under_asan(tmp_path):
    """the reader, built with AddressSanitizer + UBSan on the CPU, on the inputs that used to go wrong -- sub-byte gray
    and palette images wider than their byte stride (the per-pixel pointer ran past the row), a header that claims
    0xFFFFFFFF x 0xFFFFFFFF RGBA16 (allocation size overflow / uncaught std::length_error), and a header whose pixel count is
    merely absurd.  Every case must either decode to PIL's pixels or come back as an error string; the sanitizers abort otherwise."""
    import struct
    import subprocess
    import zlib
    PIL = pytest.importorskip("PIL.Image")
    exe = str(tmp_path / "png_asan")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        os.path.join(ROOT, "tests", "cpp", "png_reader_main.cpp"), "-o", exe, "-lz"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    rng = np.random.default_rng(11)
    g = rng.integers(0, 256, (2, 64), dtype=np.uint8)
    cases = {"bw_64x2.png": PIL.fromarray(g).convert("1"), "bw_61x7.png": PIL.fromarray(rng.integers(0, 256, (7, 61), dtype=np.uint8)).convert("1"),
             "pal4_33x5.png": PIL.fromarray(rng.integers(0, 256, (5, 33, 3), dtype=np.uint8)).quantize(16),
             "pal2_19x3.png": PIL.fromarray(rng.integers(0, 256, (3, 19, 3), dtype=np.uint8)).quantize(4),
             "pal1_70x2.png": PIL.fromarray(rng.integers(0, 256, (2, 70, 3), dtype=np.uint8)).quantize(2)}
    for name, im in cases.items():
        path = str(tmp_path / name)
        im.save(path, bits={"pal4_33x5.png": 4, "pal2_19x3.png": 2, "pal1_70x2.png": 1}.get(name, 1)) if name.startswith("pal") else im.save(path)
        out = str(tmp_path / (name + ".raw"))
        r = subprocess.run([exe, path, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, (name, r.stdout)
        rows, cols = map(int, r.stdout.split())
        want = np.asarray(PIL.open(path).convert("RGB"))[:, :, ::-1]
        assert np.array_equal(np.fromfile(out, np.uint8).reshape(rows, cols, 3), want), name

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    sig = bytes([0x89, 0x50, 0x4E, 0x47, 0x0D, 0x0A, 0x1A, 0x0A])
    for name, (w, h, depth, color) in {"huge.png": (0xFFFFFFFF, 0xFFFFFFFF, 16, 6), "wide.png": (70000, 2, 8, 2),
                                       "many.png": (60000, 60000, 8, 0), "zero.png": (0, 5, 8, 0)}.items():
        data = sig + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
        (tmp_path / name).write_bytes(data)
        r = subprocess.run([exe, str(tmp_path / name)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 3 and r.stdout.startswith("error:") and ("out of range" in r.stdout or "no image header" in r.stdout), (name, r.stdout)
    # a consistent header whose data is too short for it
    data = sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 64, 64, 1, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 10)) + chunk(b"IEND", b"")
    (tmp_path / "short.png").write_bytes(data)
    r = subprocess.run([exe, str(tmp_path / "short.png")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 3 and "inflate" in r.stdout, r.stdout


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher must start N ranks itself, as child processes, before it touches a GPU
    (it silently measured one rank).  Without a GPU every rank reports that and exits 2; the parent passes it on."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode == 2, (r.returncode, r.stderr[-500:])
        assert r.stderr.count("no GPU visible") == 2, r.stderr[-500:]


def test_run_scripts_keep_the_reference_parameters():
    """script/run_*.sh: one launcher per reference script (cpp_code/script/), same 13 parameter VALUES in the order of sfm.cpp:35-50.
    A stub in place of the driver records what it is called with."""
    import subprocess
    import tempfile
    want = {   # feature, parameter, ransac px, find init pair, calib tolerance, BA every (the reference scripts' values)
        "run_fountain_small.sh": ("S", "300", "1.0", "1", "0", "4"), "run_fountain_large.sh": ("S", "300", "1.0", "1", "0", "4"),
        "run_gerrardhall.sh": ("S", "500", "1.0", "1", "0", "5"), "run_personhall.sh": ("S", "600", "1.0", "1", "0", "5"),
        "run_southbuilding.sh": ("S", "500", "1.0", "1", "0", "5"), "run_zurich.sh": ("S", "300", "2.0", "0", "20.0", "4"),
    }
    with tempfile.TemporaryDirectory() as td:
        stub = os.path.join(td, "stub.sh")
        with open(stub, "w") as f:
            f.write('#!/bin/bash\necho "$#" "$@"\nexit 1\n')
        os.chmod(stub, 0o755)
        for name, vals in want.items():
            env = dict(os.environ, SFM_BIN=stub, SFM_DATA="/data/x", SFM_OUT=os.path.join(td, "o", "c.ply"))
            r = subprocess.run(["bash", os.path.join(ROOT, "script", name)], env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
            a = r.stdout.splitlines()[0].split()
            assert r.returncode == 1 and a[0] == "13", (name, r.stdout)
            assert tuple(a[6:12]) == vals and a[1].startswith("/data/x/") and a[5].endswith("c.ply"), (name, a)
            assert (a[4] == "none") == (name in ("run_fountain_small.sh", "run_fountain_large.sh", "run_zurich.sh"))


def test_native_jpeg_reader(tmp_path):
    """the native driver reads baseline JPEG (the COLMAP data sets of script/run_*hall.sh, run_southbuilding.sh).
    Files written by tests/jpeg_util.encode (own baseline encoder: no imaging library in the image) in 4:4:4 / 4:2:2 / 4:2:0 /
    gray, odd sizes, with and without restart intervals; the decoded BGR must equal, bit for bit, the numpy restatement of
    libjpeg's pipeline (islow IDCT, fancy upsampling, 16-bit colour tables) applied to the coefficients the encoder wrote, and
    be close to the source image.  Progressive files and truncated ones come back as error strings.  (Parity unpinned: neither
    side has been compared with libjpeg itself.)"""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import jpeg_util
    exe = os.path.join(ROOT, "bin", "sfm_native")
    if not os.path.exists(exe):
        pytest.skip("bin/sfm_native not built")
    rng = np.random.default_rng(5)

    def picture(h, w):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        base = np.stack([128 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 60 + 1.5 * xx + 0.5 * yy, 200 - 1.2 * yy + 20 * np.sin((xx + yy) / 5.0)], axis=2)
        return np.clip(base + rng.normal(0, 4, (h, w, 3)), 0, 255).astype(np.uint8)

    def dump(path):
        out = path + ".raw"
        r = subprocess.run([exe, "--dump-image", path, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            return r.returncode, r.stdout
        raw = np.fromfile(out, np.uint8)
        rows, cols = np.frombuffer(raw[:8].tobytes(), np.int32)
        return 0, raw[8:].reshape(rows, cols, 3)

    cases = [("444", 40, 56, 0, 92), ("420", 37, 53, 0, 90), ("420", 64, 48, 3, 75), ("422", 33, 70, 5, 85), ("420", 17, 9, 1, 95), ("gray", 29, 41, 0, 88),
             ("420", 8, 8, 0, 90), ("444", 1, 1, 0, 90), ("420", 2, 3, 0, 90)]
    for sub, h, w, rst, q in cases:
        img = picture(h, w)
        src = img[:, :, 1] if sub == "gray" else img
        data, info = jpeg_util.encode(src, subsampling="444" if sub == "gray" else sub, quality=q, restart=rst)
        path = str(tmp_path / f"{sub}_{h}x{w}_{rst}.jpg")
        with open(path, "wb") as f:
            f.write(data)
        rc, got = dump(path)
        assert rc == 0, (sub, h, w, got)
        want = jpeg_util.reference_decode(info)
        assert got.shape == want.shape and np.array_equal(got, want), (sub, h, w, rst, int(np.abs(got.astype(int) - want).max()))
        ref = np.stack([src] * 3, axis=2) if sub == "gray" else img[:, :, ::-1]
        if h >= 16 and w >= 16:
            mse = np.mean((got.astype(np.float64) - ref) ** 2)
            assert 10 * np.log10(255.0 ** 2 / mse) > 28.0, (sub, h, w, mse)
    data, _ = jpeg_util.encode(picture(32, 32), "420", 90)
    prog = data.replace(b"\xFF\xC0", b"\xFF\xC2", 1)
    for name, blob, msg in (("prog.jpg", prog, "progressive"), ("trunc.jpg", data[:len(data) // 3], ""), ("hdr.jpg", data[:30], "")):
        (tmp_path / name).write_bytes(blob)
        rc, out = dump(str(tmp_path / name))
        assert rc == 3 or (rc == 0 and name == "trunc.jpg"), (name, rc, out)      # a cut entropy stream decodes to padding, as libjpeg's does
        if msg:
            assert msg in out

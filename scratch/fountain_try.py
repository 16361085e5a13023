import sys, io, contextlib
sys.path.insert(0, "/root/repo")
import numpy as np
import easysfm_amd as E
z = np.load("/root/repo/tests/golden/fountain11_half_gray.npz")
K = np.array([[689.87 / 2, 0, 380.17 / 2], [0, 691.04 / 2, 251.70 / 2], [0, 0, 1]], np.float32)
ctx = E.Context(0, None)
for thr in (300, 100, 30):
    frames = []
    for i, img in enumerate(z["images"]):
        fr = E.Frame(frame_id=i, rgb_image=img); fr.K_cam = K.copy()
        with contextlib.redirect_stdout(io.StringIO()):
            E.detectFeaturesSURF(fr, thr, ctx=ctx)
        frames.append(fr)
    print("thr", thr, "features", [len(f.keypoints) for f in frames])
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            cloud, filt, graph = E.run_sfm(frames, None, "S", 1.0, True, 0.0, 4, ctx)
        print("  neighbour inliers", [len(graph[i][i-1].matches) for i in range(1, 11)], "points", len(cloud.xyz), len(filt.xyz))
        lines = [l for l in buf.getvalue().split("\n") if "initial corr" in l or "Inlier count" in l]
        print("  ", lines)
    except Exception as e:
        lines = [l for l in buf.getvalue().split("\n") if "initial corr" in l or "Inlier count" in l]
        print("  FAILED", repr(e)[:100], lines)

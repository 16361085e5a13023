# per-frame trace of bin/sfm_native (ESFM_FRAME_TRACE=1) inside bench.py's config-1 / config-3 legs
import os, sys, subprocess; sys.path.insert(0, '.')
os.environ["ESFM_FRAME_TRACE"] = "1"
import bench
_run = subprocess.run
def run(*a, **k):
    r = _run(*a, **k)
    for l in r.stdout.splitlines():
        if l.startswith("[frame") or l.startswith("stage seconds"):
            print(l[:400])
    print("--")
    return r
subprocess.run = run
for tag, f, p in (('config1', 'S', 300), ('config3', 'O', 8000)):
    r = bench.e2e_leg(tag, f, p, False)
    print(tag, 'wall', round(r['value'], 3))

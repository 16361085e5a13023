"""esfm_undistort on a 3072 x 2048 BGR image from pageable host memory (fresh output each call, kept output) and from pinned host
memory (torch.empty(pin_memory=True)): milliseconds per call.  common.hpp copy_h2d / copy_d2h: pageable buffers travel in 512-KiB
pieces, registered ones in one transfer."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import easysfm_amd as E
from easysfm_amd._lib import lib, check
ctx = E.Context(0, None)
rows, cols = 2048, 3072
img = np.random.default_rng(4400).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
k4 = np.array([2759.48, 1520.69, 2764.16, 1006.81]); d4 = np.array([-0.12, 0.03, 0.001, -0.0005])
def call(src, dst):
    check(lib().esfm_undistort(ctx.handle, C.c_void_p(src.ctypes.data), rows, cols, 3, C.c_void_p(k4.ctypes.data), C.c_void_p(d4.ctypes.data), C.c_void_p(dst.ctypes.data)))
def bench(tag, mk):
    src, dst = mk()
    call(src, dst); call(src, dst)
    ts = []
    for _ in range(10):
        src, dst = mk()
        t0 = time.perf_counter(); call(src, dst); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{tag}: median {np.median(ts):.3f} ms, min {min(ts):.3f}, max {max(ts):.3f}")
    return dst
keep = np.empty_like(img)
a = bench("pageable, fresh output buffer per call", lambda: (img, np.empty_like(img)))
b = bench("pageable, same buffers", lambda: (img, keep))
pin_in = torch.empty((rows, cols, 3), dtype=torch.uint8).pin_memory(); pin_in.numpy()[:] = img
pin_out = torch.empty((rows, cols, 3), dtype=torch.uint8).pin_memory()
c = bench("pinned (registered) buffers", lambda: (pin_in.numpy(), pin_out.numpy()))
print("same result:", bool(np.array_equal(a, b) and np.array_equal(a, c)))

"""scratch/copy_paths.py's measurement for a buffer registered with hipHostRegister (the cv::Mat case of INTEGRATION.md)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import easysfm_amd as E
from easysfm_amd._lib import lib, check
ctx = E.Context(0, None)
rows, cols = 2048, 3072
img = np.random.default_rng(4400).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
out = np.zeros_like(img)
k4 = np.array([2759.48, 1520.69, 2764.16, 1006.81]); d4 = np.array([-0.12, 0.03, 0.001, -0.0005])
def call():
    check(lib().esfm_undistort(ctx.handle, C.c_void_p(img.ctypes.data), rows, cols, 3, C.c_void_p(k4.ctypes.data), C.c_void_p(d4.ctypes.data), C.c_void_p(out.ctypes.data)))
def timeit(tag):
    call(); call()
    t0 = time.perf_counter()
    for _ in range(10): call()
    print(f"{tag}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per call")
timeit("pageable numpy buffers")
hip = C.CDLL("libamdhip64.so")
t0 = time.perf_counter()
r1 = hip.hipHostRegister(C.c_void_p(img.ctypes.data), C.c_size_t(img.nbytes), C.c_uint(0)); r2 = hip.hipHostRegister(C.c_void_p(out.ctypes.data), C.c_size_t(out.nbytes), C.c_uint(0))
print("hipHostRegister x2:", r1, r2, f"{(time.perf_counter() - t0) * 1e3:.2f} ms")
timeit("the same buffers, registered")
hip.hipHostUnregister(C.c_void_p(img.ctypes.data)); hip.hipHostUnregister(C.c_void_p(out.ctypes.data))

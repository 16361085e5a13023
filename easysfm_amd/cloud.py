"""Host-side mirror of the reference's sparse-cloud post-processing over the C ABI:
``CProceesing.SORFilter`` (cpp_code/include/cloudprocessing.hpp:24-36) and ``DataIO.writePlyFile``
(cpp_code/src/data_io.cpp:147-165).  The k-nearest-neighbour pass of the filter runs in libesfm_hip.so on the GPU; the
.ply writer is plain host I/O."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from ._lib import Context, check, default_context, lib
from .types import SparsePointCloud


def sor_filter(points, mean_k: int = 50, std_mul: float = 2.0, ctx: Optional[Context] = None) -> Tuple[np.ndarray, np.ndarray, float]:
    """esfm_sor_filter.  points: [n, stride >= 3] float32 with x, y, z first.
    Returns (keep mask [n] bool, mean k-NN distances [n] float32, threshold)."""
    ctx = ctx or default_context()
    pts = np.ascontiguousarray(points, np.float32)
    if pts.ndim != 2 or pts.shape[1] < 3:
        raise ValueError("points must be [n, >= 3]")
    n, stride = pts.shape
    md = np.zeros(max(n, 1), np.float32); keep = np.zeros(max(n, 1), np.uint8)
    n_keep = C.c_int32(0); thr = C.c_double(0.0)
    check(lib().esfm_sor_filter(ctx.handle, C.c_void_p(pts.ctypes.data), n, stride, int(mean_k), float(std_mul),
                                C.c_void_p(md.ctypes.data), C.c_void_p(keep.ctypes.data), C.byref(n_keep), C.byref(thr)))
    return keep[:n].astype(bool), md[:n], thr.value


class CProceesing:
    """Mirror of ``CProceesing<PointT>`` (cloudprocessing.hpp:20-72; the reference's spelling), SOR filter only."""

    def __init__(self, ctx: Optional[Context] = None):
        self._ctx = ctx

    def SORFilter(self, incloud: SparsePointCloud, MeanK: int = 50, std: float = 2.0) -> SparsePointCloud:
        """cloudprocessing.hpp:24-36.  Returns the filtered cloud (the reference fills ``outcloud``); survivors keep
        their input order, colours travel with the points; track ids / inlier flags are not part of a pcl cloud."""
        xyz = np.ascontiguousarray(incloud.xyz, np.float32).reshape(-1, 3)
        keep, _, _ = sor_filter(xyz, MeanK, std, self._ctx)
        out = SparsePointCloud(xyz=xyz[keep].copy())
        rgb = np.asarray(incloud.rgb)
        if rgb.shape[0] == xyz.shape[0]:
            out.rgb = rgb[keep].copy()
        print(f"apply SOR filter: [ {xyz.shape[0]} ] points before filtering, [ {int(keep.sum())} ] points after filtering.")
        return out


def _fmt(v: float) -> str:
    """operator<< of a float on a stream with precision 8 (pcl::PLYWriter::writeASCII's default): %.8g."""
    return "%.8g" % float(np.float32(v))


def write_ply(file_name: str, cloud: SparsePointCloud) -> bool:
    """DataIO::writePlyFile (data_io.cpp:147-165): width = 1, height = N (:151-152), then pcl::io::savePLYFile, i.e. an
    ASCII PLY of PointXYZRGB with the camera element PCL appends [upstream pcl/io/ply_io.cpp PLYWriter::generateHeader /
    writeASCII, restated from memory; the reference ships no example file, SURVEY.md section 8 f-3]."""
    xyz = np.asarray(cloud.xyz, np.float32).reshape(-1, 3)
    n = xyz.shape[0]
    rgb = np.asarray(cloud.rgb, np.uint8).reshape(-1, 3) if len(cloud.rgb) == n else np.zeros((n, 3), np.uint8)
    width, height = 1, n
    head = ["ply", "format ascii 1.0", "comment PCL generated", f"element vertex {n}",
            "property float x", "property float y", "property float z",
            "property uchar red", "property uchar green", "property uchar blue",
            "element camera 1",
            "property float view_px", "property float view_py", "property float view_pz",
            "property float x_axisx", "property float x_axisy", "property float x_axisz",
            "property float y_axisx", "property float y_axisy", "property float y_axisz",
            "property float z_axisx", "property float z_axisy", "property float z_axisz",
            "property float focal", "property float scalex", "property float scaley",
            "property float centerx", "property float centery",
            "property int viewportx", "property int viewporty",
            "property float k1", "property float k2", "end_header"]
    try:
        with open(file_name, "w") as f:
            f.write("\n".join(head) + "\n")
            for i in range(n):
                f.write(f"{_fmt(xyz[i, 0])} {_fmt(xyz[i, 1])} {_fmt(xyz[i, 2])} {int(rgb[i, 0])} {int(rgb[i, 1])} {int(rgb[i, 2])}\n")
            # sensor origin 0, identity orientation, no focal / scale / centre, viewport = width x height, no k1 k2
            f.write(f"0 0 0 1 0 0 0 1 0 0 0 1 0 0 0 0 0 {width} {height} 0 0\n")
    except OSError:
        print("Couldn't write file ")
        return False
    print(f"Output [ {n} ] points.\nOutput ply file done.")
    return True


def read_ply_vertices(file_name: str):
    """Minimal reader for the files write_ply produces (round-trip tests)."""
    with open(file_name) as f:
        lines = f.read().split("\n")
    n = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
    body = lines[lines.index("end_header") + 1:]
    arr = np.array([l.split() for l in body[:n]], np.float64).reshape(n, 6)
    return arr[:, :3].astype(np.float32), arr[:, 3:].astype(np.uint8), body[n]

"""Plain-data mirrors of the reference's shared types (cpp_code/include/utility.h).

Only the fields the hot path touches are kept; OpenCV/PCL/Eigen value types become numpy arrays.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, NamedTuple, Optional

import numpy as np


class DMatch(NamedTuple):
    """cv::DMatch as the reference uses it (feature_matching.cpp:90,135)."""
    queryIdx: int
    trainIdx: int
    distance: float
    imgIdx: int = 0


@dataclass
class Frame:
    """frame_t (utility.h:21-55): descriptors + track bookkeeping + pose."""
    frame_id: int = 0
    image_file_path: str = ""
    keypoints: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))   # cv::KeyPoint::pt
    descriptors: Optional[np.ndarray] = None        # CV_32F [N,64] (SURF) or CV_8U [N,32] (ORB)
    unique_pixel_ids: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    unique_pixel_has_match: np.ndarray = field(default_factory=lambda: np.zeros(0, bool))
    pose_cam: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))     # Eigen::Matrix4f
    K_cam: np.ndarray = field(default_factory=lambda: np.eye(3, dtype=np.float32))        # Eigen::Matrix3f
    rgb_image: Optional[np.ndarray] = None          # cv::Mat CV_8UC3, BGR as cv::imread delivers it (utility.h:27)

    def init_pixel_ids(self) -> None:
        """frame_t::init_pixel_ids (utility.h:47-54)."""
        n = len(self.keypoints)
        self.unique_pixel_ids = np.full(n, -1, np.int64)
        self.unique_pixel_has_match = np.zeros(n, bool)


@dataclass
class SparsePointCloud:
    """pointcloud_sparse_t (utility.h:88-102): float xyz + rgb + track ids."""
    xyz: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))   # pcl::PointXYZRGB x,y,z (float)
    rgb: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.uint8))
    unique_point_ids: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    is_inlier: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))

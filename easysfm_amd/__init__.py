"""easysfm_amd -- MI355X-native hot path of EasySFM (pairwise descriptor matching + bundle adjustment).

The compute lives in libesfm_hip.so (hand-written HIP for gfx950 behind the C ABI of include/esfm.h);
this package is the Python-side mirror of the reference's FeatureMatching / BundleAdjustment interface.
Importing the package does not touch the GPU; the shared library is loaded on first use and there is
no CPU fallback.
"""
from ._lib import BAOptions, BASummary, Context, EsfmError, ESFM_HAMMING, ESFM_L2_F32, LIB_PATH, lib  # noqa: F401
from .types import DMatch, Frame, SparsePointCloud  # noqa: F401
from .matching import (DescriptorBank, FeatureMatching, PairMatcher, knn_match_hamming, knn_match_l2,  # noqa: F401
                       match_hamming, match_l2, match_pairs_host, shard_pair_list)
from .ba import (BAProblem, BundleAdjustment, Comm, ba_solve, ba_solve_ex, ba_sweep_bytes_per_obs, default_options, line_search_next_step, reduced_plan, shard_points,  # noqa: F401
                 torch_allreduce_callback)

from .cloud import CProceesing, read_ply_vertices, sor_filter, write_ply  # noqa: F401
from .motion import (MotionEstimator, find_essential_mat, find_essential_pairs, five_point_models, pixel2cam, ransac_sample_stream, recover_pose,  # noqa: F401
                     recover_pose_pairs, solve_pnp_ransac, triangulate_pairs, triangulate_points)

from .features import detectFeaturesORB, detectFeaturesSURF, import_distort, orb_detect_and_compute, surf_detect_and_compute, undistort  # noqa: F401
from .pipeline import FramePair, match_and_verify_all_pairs, propagate_track_ids, run_sfm  # noqa: F401

__version__ = "0.1.0"

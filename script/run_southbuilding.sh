#!/bin/bash
# COLMAP south-building subset, with a lens distortion file
# Same parameter values as the reference's cpp_code/script/run_southbuilding.sh; the data set location comes from SFM_DATA
# (the reference hard-codes the author's disk).  FEATURE=O selects ORB (then FEATURE_PARAM is the feature budget, e.g. 8000).
. "$(dirname "$0")/common.sh"
D=${SFM_DATA:-sfm_data/south-building}
IMG_DIR=$D/test_img  IMG_LIST=$D/sub_image_list.txt  K_FILE=$D/Calibration/K.txt  DISTORT_FILE=$D/Calibration/distort.txt
OUT_PLY=${SFM_OUT:-output/sfm_sparse_point_cloud_southbuilding.ply}
FEATURE=${FEATURE:-S}  FEATURE_PARAM=${FEATURE_PARAM:-500}  RANSAC_PX=1.0  FIND_INIT_PAIR=1  CALIB_TOL=0  BA_EVERY=5
run_sfm

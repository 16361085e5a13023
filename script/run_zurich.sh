#!/bin/bash
# Zurich urban MAV subset: first two frames initialise, intrinsics free within +-20 (bounded BA with line search)
# Same parameter values as the reference's cpp_code/script/run_zurich.sh; the data set location comes from SFM_DATA
# (the reference hard-codes the author's disk).  FEATURE=O selects ORB (then FEATURE_PARAM is the feature budget, e.g. 8000).
. "$(dirname "$0")/common.sh"
D=${SFM_DATA:-sfm_data/zurich_urban}
IMG_DIR=$D/test_img  IMG_LIST=$D/sub_image_list.txt  K_FILE=$D/Calibration/K.txt  DISTORT_FILE=none
OUT_PLY=${SFM_OUT:-output/sfm_sparse_point_cloud_zurichurban.ply}
FEATURE=${FEATURE:-S}  FEATURE_PARAM=${FEATURE_PARAM:-300}  RANSAC_PX=2.0  FIND_INIT_PAIR=0  CALIB_TOL=20.0  BA_EVERY=4
run_sfm

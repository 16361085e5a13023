#!/bin/bash
# fountain-P11 at 768 x 512 (the 11 images shipped with the reference as test_data/images_25)
# Same parameter values as the reference's cpp_code/script/run_fountain_small.sh; the data set location comes from SFM_DATA
# (the reference hard-codes the author's disk).  FEATURE=O selects ORB (then FEATURE_PARAM is the feature budget, e.g. 8000).
. "$(dirname "$0")/common.sh"
D=${SFM_DATA:-test_data}
IMG_DIR=$D/images_25  IMG_LIST=$D/image_list.txt  K_FILE=$D/k_25/K.txt  DISTORT_FILE=none
OUT_PLY=${SFM_OUT:-output/sfm_sparse_point_cloud_fountain.ply}
FEATURE=${FEATURE:-S}  FEATURE_PARAM=${FEATURE_PARAM:-300}  RANSAC_PX=1.0  FIND_INIT_PAIR=1  CALIB_TOL=0  BA_EVERY=4
run_sfm

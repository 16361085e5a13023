#!/bin/bash
# Shared launcher of the script/run_*.sh files: the reference's thirteen positional arguments, in the order of
# cpp_code/test/sfm.cpp:35-50 (= cpp_code/script/run_fountain_small.sh:22-24), handed to one of this repo's two drivers.
#   SFM_BIN=bin/sfm_native (default; C++ host over libesfm_hip.so)   or   SFM_BIN=bin/sfm (Python host, same arguments)
# Run from the repository root, e.g.  SFM_DATA=/data/fountain script/run_fountain_small.sh
# Exit status 1 means success, as in the reference (sfm.cpp:339).
run_sfm() {
    local bin=${SFM_BIN:-bin/sfm_native}
    mkdir -p "$(dirname "$OUT_PLY")"
    "$bin" "$IMG_DIR" "$IMG_LIST" "$K_FILE" "${DISTORT_FILE:-none}" "$OUT_PLY" \
           "${FEATURE:-S}" "${FEATURE_PARAM:-300}" "${RANSAC_PX:-1.0}" "${FIND_INIT_PAIR:-1}" "${CALIB_TOL:-0}" "${BA_EVERY:-4}" \
           "${VIEWER:-0}" "${SPHERES:-0}"
    local rc=$?
    [ $rc -eq 1 ] && echo "sfm finished (status 1 = the reference's success code): $OUT_PLY"
    return $rc
}

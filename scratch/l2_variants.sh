#!/bin/bash
# times l2_knn_bf16_kernel builds under scratch/variants on M-SURF-4k
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== $v"; ESFM_LIB=$GRAFT_REPO_ROOT/scratch/variants/libesfm_$v.so python3 scratch/l2_time.py 2>&1 | tail -1
done

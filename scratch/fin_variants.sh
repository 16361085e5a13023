#!/bin/bash
# finish-kernel slice counts on M-SURF-4k
cd $GRAFT_REPO_ROOT
for s in 8 4 2 1; do echo "== S=$s"; ESFM_FIN_SLICES=$s python3 scratch/l2_time.py 2>&1 | tail -1; done

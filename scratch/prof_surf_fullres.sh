#!/bin/bash
# kernel trace of SURF / ORB detection at 3072 x 2048 (scratch/surf_fullres_time.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_surf_fullres; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python3 scratch/surf_fullres_time.py > $O/log.txt 2>&1
python3 tools/rocprof_csv_summary.py $O/t | head -24 | cut -c1-170
find $O -name "*.csv" -size +2M -delete

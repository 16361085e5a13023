#!/bin/bash
# kernel trace of ORB detect + describe on the 11 fountain images (scratch/orb_first_touch.py: two passes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_orb_detect; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python3 scratch/orb_first_touch.py > $O/log.txt 2>&1
python3 tools/rocprof_csv_summary.py $O/t | head -16 | cut -c1-170
grep pass $O/log.txt
find $O -name "*.csv" -size +2M -delete

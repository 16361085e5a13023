#!/bin/bash
# kernel trace of SURF detect + describe on the bench leg's image (scratch/surf_time.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_surf; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python3 scratch/surf_time.py > $O/log.txt 2>&1
python3 tools/rocprof_csv_summary.py $O/t | head -16 | cut -c1-170
tail -3 $O/log.txt
find $O -name "*.csv" -size +2M -delete

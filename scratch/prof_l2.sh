#!/bin/bash
# kernel trace of scratch/l2_time.py: true kernel durations of the matcher's launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/prof_l2
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python3 scratch/l2_time.py > $O/log.txt 2>&1
python3 tools/rocprof_csv_summary.py $O/t | head -20
tail -1 $O/log.txt
find $O -name "*.csv" -size +2M -delete

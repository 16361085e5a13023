#!/bin/bash
# kernel trace of small BA calls (scratch/ba_small_time.py): which launches make up an LM iteration of a dozen-camera problem
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_ba_small; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python3 scratch/ba_small_time.py > $O/log.txt 2>&1
python3 tools/rocprof_csv_summary.py $O/t | head -30 | cut -c1-170
tail -3 $O/log.txt
find $O -name "*.csv" -size +2M -delete

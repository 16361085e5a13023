#!/bin/bash
# round 5: the whole GPU suite after the structure-aware solve went in, then the per-kernel split of BA-512 (sparse / dense)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/prof_r05a
rm -rf $O; mkdir -p $O
python3 -m pytest tests -m gpu -x -q --durations=15 > $O/gputests.log 2>&1
tail -25 $O/gputests.log
rocprofv3 --kernel-trace --output-format csv -d $O/ba512 -o ba512 -- python3 scratch/ba512.py > $O/ba512_prof.log 2>&1
python3 tools/rocprof_csv_summary.py $O/ba512 > $O/r05_ba512_kernel_stats.txt
grep -E "^iters|^solve|^schur|^linearize" $O/ba512_prof.log >> $O/r05_ba512_kernel_stats.txt
find $O -name "*.csv" -size +2M -delete
head -40 $O/r05_ba512_kernel_stats.txt

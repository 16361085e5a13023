#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ba512prof
rocprofv3 --kernel-trace --stats -d gpurun_out/ba512prof -- python3 scratch/ba512.py > gpurun_out/ba512prof/run.log 2>&1
db=$(find gpurun_out/ba512prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db | head -24 | cut -c1-175
find gpurun_out/ba512prof -name "*.db" -delete

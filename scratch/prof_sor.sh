#!/bin/bash
# rocprofv3 kernel trace of scratch/sor_bench_cloud.py: the SOR pass's kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-sorprof}
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o r -- python3 scratch/sor_bench_cloud.py > $out/run.log 2>&1
tail -1 $out/run.log
python3 - $out <<'PY'
import csv, glob, collections, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
d = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    d[row['Kernel_Name'].split('(')[0][:70]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k:72s} {len(v):5d} calls  avg {sum(v)/len(v):9.2f} us  min {min(v):9.2f}  max {max(v):9.2f}')
PY

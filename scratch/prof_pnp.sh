#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/pnpprof
rm -rf $out; mkdir -p $out
python3 scratch/pnp_time.py
rocprofv3 --kernel-trace --output-format csv -d $out -o r -- python3 scratch/pnp_time.py > $out/run.log 2>&1
tail -1 $out/run.log
python3 - $out <<'PY'
import csv, glob, collections, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
d = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    d[row['Kernel_Name'].split('(')[0][:60]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k:50s} {len(v):6d} calls  total {sum(v):10.1f} us  avg {sum(v)/len(v):9.2f}  min {min(v):9.2f}  max {max(v):9.2f}')
PY

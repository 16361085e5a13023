#!/bin/bash
# rocprofv3 kernel trace of scratch/ransac_bench.py: essential_* kernel times
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-ransacprof}
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o r -- python3 scratch/ransac_bench.py > $out/run.log 2>&1
grep pairs= $out/run.log
python3 - $out <<'PY'
import csv, glob, collections, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
d = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    d[row['Kernel_Name'].split('(')[0][:60]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if 'essential' in k or 'pose' in k: print(f'{k:50s} {len(v):6d} calls  total {sum(v):10.1f} us  avg {sum(v)/len(v):9.2f}  min {min(v):9.2f}  max {max(v):9.2f}')
PY

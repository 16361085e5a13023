#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-ba25prof}
mkdir -p $out
python3 scratch/ba25.py 2>&1 | tail -1
rocprofv3 --kernel-trace --output-format csv -d $out -o ba25 -- python3 scratch/ba25.py > $out/run.log 2>&1
python3 - $out <<'PY'
import csv, glob, collections, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
d = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    d[row['Kernel_Name'].split('(')[0][:60]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
print(f'# kernel | calls | total_us | avg_us | min | max | pct')
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k:60s} {len(v):6d} {sum(v):12.1f} {sum(v)/len(v):9.2f} {min(v):9.2f} {max(v):9.2f} {100*sum(v)/tot:6.2f}')
PY

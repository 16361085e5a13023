#!/bin/bash
# rocprofv3 kernel trace + stats over scratch/l2_time.py; $1 = tag
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-trace_l2}
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 scratch/l2_time.py > $out/log.txt 2>&1
python3 tools/rocprof_csv_summary.py $out 2>/dev/null | head -30 || true
ls $out
python3 - $out <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        d[row['Kernel_Name'][:70]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(f'{k:72s} n {len(v):4d} avg {sum(v)/len(v):9.1f} us  min {min(v):9.1f}  max {max(v):9.1f}')
PY

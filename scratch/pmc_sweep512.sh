#!/bin/bash
# HBM traffic and issue counters of the BA-512 Jacobian sweep, per dispatch (the first launches are the two-pass form)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_sweep512
rm -rf $out; mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o fetch -- python3 scratch/ba512.py > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out -o write -- python3 scratch/ba512.py > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o sq -- python3 scratch/ba512.py > $out/sq.log 2>&1
python3 - $out <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        if 'ba_linearize' in row['Kernel_Name']:
            per[row['Counter_Name']][int(row['Dispatch_Id'])] += float(row['Counter_Value'])
    for c, dd in sorted(per.items()):
        vals = [dd[k] for k in sorted(dd)]
        print(f'{c:22s} launches {len(vals)}  first {vals[0]:.4g}  median {sorted(vals)[len(vals)//2]:.4g}  max {max(vals):.4g}  min {min(vals):.4g}')
PY

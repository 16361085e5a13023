#!/bin/bash
# issue / wait counters of the BA-512 Jacobian sweep for the library in $1 (default in-tree): bash scratch/pmc_sweep_r06.sh [lib.so] tag
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${2:-intree}
[ -n "$1" ] && [ "$1" != "-" ] && export ESFM_LIB=$GRAFT_REPO_ROOT/$1
out=gpurun_out/pmc_sweep_r06_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o sq -- python3 scratch/ba512.py > $out/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out -o sq2 -- python3 scratch/ba512.py > $out/sq2.log 2>&1
python3 - $out $tag <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        if 'ba_linearize' in row['Kernel_Name']:
            per[row['Counter_Name']][int(row['Dispatch_Id'])] += float(row['Counter_Value'])
    for c, dd in sorted(per.items()):
        vals = [dd[k] for k in sorted(dd)]
        print(f'{sys.argv[2]:10s} {c:22s} launches {len(vals)}  median {sorted(vals)[len(vals)//2]:.5g}  min {min(vals):.5g}  max {max(vals):.5g}')
PY
find $out -name "*.csv" -size +1M -delete

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_schur
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $out -o sq -- python3 scratch/ba512.py > $out/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out -o inst -- python3 scratch/ba512.py > $out/inst.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out -o fetch -- python3 scratch/ba512.py > $out/fetch.log 2>&1
python3 - $out <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:34]
        acc[(k, row['Counter_Name'])] += float(row['Counter_Value'])
        disp[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
    for (k, c), v in sorted(acc.items()):
        if 'schur_mfma' in k or 'point_prep' in k: print(f'{k:36s} {c:24s} per-launch {v / len(disp[(k, c)]):16.1f}  launches {len(disp[(k, c)])}')
PY

#!/bin/bash
# PMC passes (SQ group, instruction group) over scratch/l2_time.py; $1 = output tag, ESFM_LIB may select a variant library
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
out=gpurun_out/${1:-pmc_x1}
mkdir -p $out
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o sq -- python3 scratch/l2_time.py > $out/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out -o inst -- python3 scratch/l2_time.py > $out/inst.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_LEVEL_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $out -o occ -- python3 scratch/l2_time.py > $out/occ.log 2>&1
python3 - $out <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:40]
        acc[(k, row['Counter_Name'])] += float(row['Counter_Value'])
        disp[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
    for (k, c), v in sorted(acc.items()):
        if 'l2_knn' in k: print(f'{k:42s} {c:28s} per-launch {v / len(disp[(k, c)]):16.1f}  launches {len(disp[(k, c)])}')
for f in sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)):
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        d[row['Kernel_Name'][:40]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    for k, v in d.items():
        if 'l2_knn' in k: print(f'{f.split("/")[-1]:30s} {k:42s} avg {sum(v)/len(v):9.1f} us  n {len(v)}')
PY

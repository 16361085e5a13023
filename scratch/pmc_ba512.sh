#!/bin/bash
# PMC passes over the BA-25 loop; prints per-launch averages for kernels matching $2 (default schur_lds)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
out=gpurun_out/${1:-pmc_ba512}
pat=${2:-schur_lds}
mkdir -p $out
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -o sq -- python3 scratch/ba512.py > $out/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out -o inst -- python3 scratch/ba512.py > $out/inst.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d $out -o lds -- python3 scratch/ba512.py > $out/lds.log 2>&1
python3 - $out $pat <<'PY'
import csv, glob, collections, sys
pat = sys.argv[2]
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:48]
        acc[(k, row['Counter_Name'])] += float(row['Counter_Value'])
        disp[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
    for (k, c), v in sorted(acc.items()):
        if pat in k: print(f'{k:50s} {c:28s} per-launch {v / len(disp[(k, c)]):16.1f}  launches {len(disp[(k, c)])}')
PY

#!/bin/bash
# PMC passes over the matching leg of bench.py (separate passes per counter group; kernel-trace only, as gpurun requires)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/pmc3
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc3 -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-ba --no-cpu-baseline > gpurun_out/pmc3/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc3 -o write -- python3 bench.py --steps 2 --warmup 1 --no-ba --no-cpu-baseline > gpurun_out/pmc3/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc3 -o sq -- python3 bench.py --steps 2 --warmup 1 --no-ba --no-cpu-baseline > gpurun_out/pmc3/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc3 -o inst -- python3 bench.py --steps 2 --warmup 1 --no-ba --no-cpu-baseline > gpurun_out/pmc3/inst.log 2>&1
find gpurun_out/pmc3 -name "*counter_collection.csv" | head
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc3/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    disp = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        acc[(k, row['Counter_Name'])][0] += float(row['Counter_Value'])
        disp[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
    print('==', f)
    for (k, c), (v, _) in sorted(acc.items()):
        n = len(disp[(k, c)])
        if 'l2_knn' in k or 'split' in k or 'exact_scan' in k:
            print(f'{k:62s} {c:28s} per-launch {v / n:16.1f}  launches {n}')
PY

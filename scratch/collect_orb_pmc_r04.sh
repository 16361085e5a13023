#!/bin/bash
# PMC counters of the ORB matcher (hamming_fp4_kernel since round 4) on M-ORB-4k: one --pmc group per run, kernel trace only
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_orb
rm -rf $O; mkdir -p $O
M="python3 scratch/orb_time.py"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- $M > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_inst -o inst -- $M > $O/pmc_inst.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o fetch -- $M > $O/pmc_fetch.log 2>&1
for g in sq inst fetch; do python3 tools/rocprof_csv_summary.py $O/pmc_$g hamming_ | grep -v "^# kernel trace" | grep -E "^# counters|hamming_fp4|hamming_knn|hamming_expand"; done > $O/r04_pmc_hamming_fp4.txt
find $O -name "*.csv" -size +2M -delete
cat $O/r04_pmc_hamming_fp4.txt | cut -c1-170

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r06_fin
rm -rf $O; mkdir -p $O
M="python3 scratch/hard_match_only.py"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/inst -o inst -- $M > $O/inst.log 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -o sq -- $M > $O/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o fetch -- $M > $O/fetch.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/tcc -o tcc -- $M > $O/tcc.log 2>&1
for g in inst sq fetch tcc; do python3 tools/rocprof_csv_summary.py $O/$g l2_ | grep -E "^# counters|l2_finish|l2_knn_bf16x1"; done | tee $O/r06_pmc_l2_finish_hard.txt
tail -3 $O/tcc.log
find $O -name "*.csv" -size +2M -delete

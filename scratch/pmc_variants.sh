#!/bin/bash
# executed-instruction counters of l2_knn_bf16x1_kernel for the variant libraries named on the command line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for v in "$@"; do
  out=gpurun_out/pmcv_$v; rm -rf $out; mkdir -p $out
  export ESFM_LIB=$R/scratch/variants/libesfm_$v.so
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out -o inst -- python3 scratch/l2_time.py > $out/inst.log 2>&1
  echo "== $v"
  python3 tools/rocprof_csv_summary.py $out l2_knn_bf16x1 | grep -E "l2_knn_bf16x1" | grep -v "^#" | awk '{print $(NF-3), $(NF-2)}'
  find $out -name "*.csv" -size +1M -delete
done

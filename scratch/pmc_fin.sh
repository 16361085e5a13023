#!/bin/bash
# cache counters of l2_finish_kernel on M-SURF-4k
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
out=gpurun_out/pmc_fin; rm -rf $out; mkdir -p $out
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $out -o a -- python3 scratch/l2_time.py > $out/a.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_BUFFER_WAVEFRONTS_sum --kernel-trace --output-format csv -d $out -o b -- python3 scratch/l2_time.py > $out/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out -o c -- python3 scratch/l2_time.py > $out/c.log 2>&1
python3 tools/rocprof_csv_summary.py $out l2_finish | grep -v "^#" | grep l2_finish | awk '{print $(NF-3), $(NF-2)}'
tail -2 $out/a.log | head -1
find $out -name "*.csv" -size +1M -delete

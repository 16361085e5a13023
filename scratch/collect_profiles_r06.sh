#!/bin/bash
# round-6 profile artefacts: the bench line, kernel stats of the default bench and of the config-4/5 legs, PMC passes of the dominant
# kernels (one --pmc group per run; --no-e2e: the config-1/3 legs start bash + bin/sfm_native under the profiler's preload for nothing),
# BA-25 / BA-512 kernel stats (BA-512 now through the structure-aware solve), the hard workload's finish-kernel counters.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/prof_r06
rm -rf $O; mkdir -p $O
python3 bench.py > $O/r06_bench_line.json 2> $O/bench_stderr.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --no-cpu-baseline --no-config45 --no-e2e > $O/bench_prof.log 2>&1
python3 tools/rocprof_csv_summary.py $O/bench > $O/r06_bench_kernel_stats.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench45 -o bench45 -- python3 bench.py --no-cpu-baseline --no-e2e --steps 2 --warmup 1 > $O/bench45_prof.log 2>&1
python3 tools/rocprof_csv_summary.py $O/bench45 > $O/r06_bench_config45_kernel_stats.txt
M="python3 bench.py --steps 2 --warmup 1 --no-ba --no-cpu-baseline --no-config45 --no-e2e"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o fetch -- $M > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o write -- $M > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- $M > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/pmc_inst -o inst -- $M > $O/pmc_inst.log 2>&1
for g in fetch write sq inst; do python3 tools/rocprof_csv_summary.py $O/pmc_$g l2_ | grep -v "^# kernel trace" | grep -E "^# counters|l2_knn|l2_finish|l2_split"; done > $O/r06_pmc_l2_knn_bf16x1.txt
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_ba_fetch -o fetch -- $B > $O/pmc_ba_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_ba_write -o write -- $B > $O/pmc_ba_write.log 2>&1
for g in ba_fetch ba_write; do python3 tools/rocprof_csv_summary.py $O/pmc_$g "" | grep -E "^# counters|ba_linearize|ba_schur|ba_chol_small|ba_backsub|chol_sparse"; done > $O/r06_pmc_ba.txt
python3 scratch/ba_time.py both > $O/r06_ba_time.txt 2>&1
python3 scratch/ba512_time.py >> $O/r06_ba_time.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/ba25 -o ba25 -- python3 scratch/ba25.py > $O/ba25_prof.log 2>&1
python3 tools/rocprof_csv_summary.py $O/ba25 > $O/r06_ba25_kernel_stats.txt
rocprofv3 --kernel-trace --output-format csv -d $O/ba512 -o ba512 -- python3 scratch/ba512.py > $O/ba512_prof.log 2>&1
python3 tools/rocprof_csv_summary.py $O/ba512 > $O/r06_ba512_kernel_stats.txt
grep -E "^iters|^solve|^schur|^linearize" $O/ba512_prof.log >> $O/r06_ba512_kernel_stats.txt
# keep the merge small: drop the raw traces, keep summaries and logs
ls -la $O | head -40
tail -3 $O/r06_pmc_l2_knn_bf16x1.txt

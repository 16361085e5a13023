#!/bin/bash
# full GPU checkpoint: tests, smoke, bench line, rocprof kernel stats of the bench
mkdir -p gpurun_out/ck
python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/ck/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/ck/smoke.txt 2>&1
python bench.py > gpurun_out/ck/bench.json 2> gpurun_out/ck/bench.err
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/ck/prof
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/ck/prof -- python3 bench.py --no-cpu-baseline > gpurun_out/ck/prof_bench.json 2> gpurun_out/ck/prof.err
db=$(find gpurun_out/ck/prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py $db > gpurun_out/ck/kernel_stats.txt
find gpurun_out/ck/prof -name "*.db" -delete
tail -3 gpurun_out/ck/pytest.txt; cat gpurun_out/ck/smoke.txt | tail -2; cat gpurun_out/ck/bench.json | cut -c1-600

#!/bin/bash
# ba_cost_kernel / ba_cam_rt_kernel times on BA-512 for the in-tree library and a variant: bash scratch/prof_cost_ab.sh variant.so
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for tag in intree variant; do
  if [ $tag = variant ]; then export ESFM_LIB=$GRAFT_REPO_ROOT/$1; fi
  out=gpurun_out/prof_cost_$tag; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 scratch/ba512.py > $out/log.txt 2>&1
  echo "== $tag"; grep -E "^iters" $out/log.txt
  python3 tools/rocprof_csv_summary.py $out | grep -E "ba_cost_kernel|ba_cam_rt|ba_backsub|ba_linearize" | cut -c1-60,110-180
  find $out -name "*.csv" -size +1M -delete
done

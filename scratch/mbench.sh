#!/bin/bash
# quick matcher numbers: value, ms/step, knn ms, rescan ms
python bench.py --no-ba --no-config45 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['rescan_kernel_avg_ms'],4), d['verified_vs_oracle'])
"

#!/bin/bash
# two PROCESSES on one GPU at the same time, per sort mode: does a bounded look-back spin expire (GS_ERR_SORT_TIMEOUT)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for mode in full visible; do
  for k in 1 2; do
    ( timeout 200 python bench.py --steps 30 --warmup 5 --repeats 3 --sort-mode $mode --cpu-baseline off --pmc off > gpurun_out/r06_two_$mode$k.json 2> gpurun_out/r06_two_$mode$k.err; echo "$mode proc $k rc=$?" ) &
  done
  wait
  for k in 1 2; do grep -h "GsError\|gs_error" gpurun_out/r06_two_$mode$k.err | tail -1; python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/r06_two_$mode$k.json') if l.startswith('{')][-1]); print('$mode', $k, d['ms_per_step'])
except Exception as e: print('$mode', $k, 'no line')
PY
  done
done

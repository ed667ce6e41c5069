#!/bin/bash
# round 6, call 12: dependency-ordered partition assignment (static first round + one counter): sort / draw suites, timing vs the measurement call, two processes on one GPU
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sort.py tests/test_gpu_draw.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4
for cfg in C2 C4; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --repeats 3 --sort-mode both --cpu-baseline off --pmc off > gpurun_out/r06_dep_$cfg.json 2>/dev/null
  GSPLAT_SHARED_GPU=1 timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --repeats 3 --sort-mode full --cpu-baseline off --pmc off > gpurun_out/r06_dep_${cfg}_shared.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_dep_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    print(f.split('/')[-1], {m:(x['ms_per_step'], x['stages_ms']['sort'], x['stages_ms']['bin'], x['stages_ms']['pair_sort'], x['onesweep_depth_kernel_ms']) for m,x in d['modes'].items()})
PY
for sh in 0 1; do
  for k in 1 2; do ( GSPLAT_SHARED_GPU=$sh timeout 200 python bench.py --steps 30 --warmup 5 --repeats 3 --sort-mode full --cpu-baseline off --pmc off > gpurun_out/r06_two2_$sh$k.json 2> gpurun_out/r06_two2_$sh$k.err; echo "shared=$sh proc $k rc=$?" ) & done; wait
  grep -h "gs_error" gpurun_out/r06_two2_$sh*.err | tail -2
done

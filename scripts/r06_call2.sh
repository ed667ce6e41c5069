#!/bin/bash
# round 6, call 2: 9-bit x 3 window passes against 8-bit x 4 at C4 (V = 16.8 M) and C3, same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in C4 C3; do
for w in 1 0; do
  GSPLAT_VIS_WIDE=$w timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 --sort-mode visible --cpu-baseline off --pmc off --repeats 3 > gpurun_out/r06_c2_${cfg}_wide$w.json 2> gpurun_out/r06_c2_${cfg}_wide$w.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_c2_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); m=d['modes']['visible']
        print(f, m['ms_per_step'], m['stages_ms'], m['onesweep_depth_kernel_ms'], d['config']['visible_splats'])
    except Exception as e: print(f, 'ERR', e)
PY

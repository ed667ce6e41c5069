#!/bin/bash
# experiment: cap the blend's workgroups per CU (dynamic LDS padding) so that the other frame's kernels always find wave slots
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for pad in 0 17000 30000 56000; do
  GSPLAT_BLEND_PAD_LDS=$pad timeout 300 python bench.py --steps 20 --warmup 5 --sort-mode visible_in_flight --cpu-baseline off --pmc off --repeats 3 > gpurun_out/r06_pad_$pad.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06_pad_$pad.json') if l.startswith('{')][-1])
print($pad, {m:(x['ms_per_step'], x.get('stages_ms',{}).get('blend')) for m,x in d['modes'].items()})
PY
done

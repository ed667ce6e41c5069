#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
GSPLAT_VIS_DEBUG=1 timeout 300 python bench.py --steps 4 --warmup 3 --sort-mode visible --repeats 1 --cpu-baseline off --pmc off > gpurun_out/r06_c3_dbg.json 2> gpurun_out/r06_c3_dbg.err
grep gsplat gpurun_out/r06_c3_dbg.err | head -12
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_c3_bench.json 2> gpurun_out/r06_c3_bench.err
tail -c 800 gpurun_out/r06_c3_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_c3_bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['config']['sort_mode'], d['config']['headline_reason'])
for m,x in d['modes'].items(): print(m, x['ms_per_step'], x['regions_ms_per_step'], x.get('stages_ms'), x.get('onesweep_depth_kernel_ms'))
print(d['sort_mode_cross_check'])
print(d['parity_vs_oracle'].get('visible_in_flight'))
print(d['roofline_streaming']['launches_per_frame'])
PY

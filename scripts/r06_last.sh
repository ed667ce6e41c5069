#!/bin/bash
# round 6: the last sanity call on the tree as committed: smoke(), the quick GPU tests of the newest code, the driver's command
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_random_parity.py tests/test_gpu_lanes_fuzz.py tests/test_gpu_draw.py -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_last_driver_cmd.json 2> gpurun_out/r06_last_driver_cmd.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06_last_driver_cmd.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['config']['sort_mode'], {m: x['ms_per_step'] for m, x in d['modes'].items()}, (d.get('end_of_orbit_check') or {}).get('ok'), d['roofline']['frac'])
PY

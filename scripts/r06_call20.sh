#!/bin/bash
# round 6, call 20: a target's pixels double-buffered while its context has lanes: tests, then the bench with one target per view (library), two in rotation, host-held renderers
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
timeout 1200 python -m pytest tests/test_gpu_vissort.py tests/test_gpu_draw.py tests/test_scene_depth_and_debug.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -4
for v in "library 1" "library 2" "host 1" "library 1" "library 2" "host 1"; do
  set -- $v
  timeout 600 python bench.py --steps 20 --warmup 5 --repeats 3 --sort-mode visible_in_flight --in-flight-impl $1 --in-flight-targets $2 --cpu-baseline off --pmc off 2>gpurun_out/r06_lib_$1$2.err | grep '^{' > gpurun_out/r06_lib_$1$2.json
  python - $1$2 <<'PY'
import json, sys
try:
    d = json.loads(open(f'gpurun_out/r06_lib_{sys.argv[1]}.json').read())
    print(sys.argv[1], {m: (x["ms_per_step"], x.get("regions_ms_per_step")) for m, x in d["modes"].items()}, d["sort_mode_cross_check"]["in_flight"]["ok"])
except Exception as e:
    print(sys.argv[1], 'no line', e); print(open(f'gpurun_out/r06_lib_{sys.argv[1]}.err').read()[-1500:])
PY
done

#!/bin/bash
# round 6, call 17: frames in flight inside the library (gs_renderer_set_frames_in_flight): tests, then the bench with the lanes in the library and with the host's own
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_vissort.py -x -q -m gpu -k "library or frames_dealt or views_in_flight" 2>&1 | tail -15
for impl in library host library host; do
  timeout 600 python bench.py --steps 20 --warmup 5 --repeats 3 --sort-mode visible_in_flight --in-flight-impl $impl --cpu-baseline off --pmc off 2>gpurun_out/r06_lib_$impl.err | grep '^{' > gpurun_out/r06_lib_$impl.json
  python - $impl <<'PY'
import json, sys
try:
    d = json.loads(open(f'gpurun_out/r06_lib_{sys.argv[1]}.json').read())
    print(sys.argv[1], {m: (x["ms_per_step"], x.get("regions_ms_per_step")) for m, x in d["modes"].items()}, d["sort_mode_cross_check"] and d["sort_mode_cross_check"].get("ok"), d["config"]["sort_mode"])
except Exception as e:
    print(sys.argv[1], 'no line', e); print(open(f'gpurun_out/r06_lib_{sys.argv[1]}.err').read()[-1500:])
PY
done

#!/bin/bash
# round 6, call 30: hardware queues.  The runtime maps HIP streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; a renderer with two lanes holds six streams
# (three contexts x main + second queue).  Frames in flight at C2 with 2 / 4 (default) / 8 hardware queues, alternating.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
run() {
  local name=$1; shift
  env "$@" timeout 600 python bench.py --config C2 --steps 50 --warmup 10 --repeats 5 --sort-mode visible_in_flight --cpu-baseline off --pmc off > gpurun_out/r06_hwq_$name.json 2> gpurun_out/r06_hwq_$name.err
  python - $name <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/r06_hwq_{sys.argv[1]}.json') if l.startswith('{')][-1])
    m = d["modes"]["visible_in_flight"]
    print(sys.argv[1], m["ms_per_step"], m.get("regions_ms_per_step"))
except Exception as e:
    print(sys.argv[1], 'no line', e); print(open(f'gpurun_out/r06_hwq_{sys.argv[1]}.err').read()[-800:])
PY
}
for rep in 1 2; do
  run q4_$rep GSPLAT_X=0
  run q8_$rep GPU_MAX_HW_QUEUES=8
  run q2_$rep GPU_MAX_HW_QUEUES=2
  run q16_$rep GPU_MAX_HW_QUEUES=16
done

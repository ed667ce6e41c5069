#!/bin/bash
# round 6, call 25: bounded experiment -- a lane's blend on a second queue of its context at the LOWEST queue priority (GSPLAT_BLEND_AUX=1, GSPLAT_PRIO=<main><aux>),
# so that wave slots freed by finished tiles go to the other frame's latency-bound kernels first.  Frames in flight at C2 / C3 / C5, alternating, + the lanes tests.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
run() {  # name cfg env...
  local name=$1 cfg=$2; shift 2
  env "$@" timeout 600 python bench.py --config $cfg --steps 50 --warmup 10 --repeats 5 --sort-mode visible_in_flight --cpu-baseline off --pmc off > gpurun_out/r06_prio_${cfg}_$name.json 2> gpurun_out/r06_prio_${cfg}_$name.err
  python - $cfg $name <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/r06_prio_{sys.argv[1]}_{sys.argv[2]}.json') if l.startswith('{')][-1])
    m = d["modes"]["visible_in_flight"]
    print(sys.argv[1], sys.argv[2], m["ms_per_step"], m.get("regions_ms_per_step"))
except Exception as e:
    print(sys.argv[1], sys.argv[2], 'no line', e); print(open(f'gpurun_out/r06_prio_{sys.argv[1]}_{sys.argv[2]}.err').read()[-800:])
PY
}
for rep in 1 2; do
  run base$rep C2 GSPLAT_X=0
  run aux$rep C2 GSPLAT_BLEND_AUX=1
  run auxlow$rep C2 GSPLAT_BLEND_AUX=1 GSPLAT_PRIO=01
  run hilow$rep C2 GSPLAT_BLEND_AUX=1 GSPLAT_PRIO=11
done
for cfg in C3 C5; do
  run base1 $cfg GSPLAT_X=0
  run auxlow1 $cfg GSPLAT_BLEND_AUX=1 GSPLAT_PRIO=01
  run hilow1 $cfg GSPLAT_BLEND_AUX=1 GSPLAT_PRIO=11
done
GSPLAT_BLEND_AUX=1 GSPLAT_PRIO=11 timeout 900 python -m pytest tests/test_gpu_lanes_fuzz.py tests/test_gpu_vissort.py -q -m gpu -k "lanes or flight or random" 2>&1 | tail -3

#!/bin/bash
# round 6, call 26: the other direction of call 25 -- a lane's blend on a second queue at the HIGHEST priority (GSPLAT_PRIO=02), and with the lanes' own queues at the lowest (22)
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
  run auxhigh$rep C2 GSPLAT_BLEND_AUX=1 GSPLAT_PRIO=02
  run lowhigh$rep C2 GSPLAT_BLEND_AUX=1 GSPLAT_PRIO=22
done
run base1 C3 GSPLAT_X=0
run auxhigh1 C3 GSPLAT_BLEND_AUX=1 GSPLAT_PRIO=02
run lowhigh1 C3 GSPLAT_BLEND_AUX=1 GSPLAT_PRIO=22

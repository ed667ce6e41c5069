#!/bin/bash
# round 6, call 4: frames / views in flight on 2..4 renderers (contexts = streams) over one copy of the asset, against one at a time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { cfg=$1; nf=$2; shift 2; timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 --cpu-baseline off --pmc off --repeats 3 --in-flight $nf "$@" > gpurun_out/r06_c4_${cfg}_f$nf.json 2> gpurun_out/r06_c4_${cfg}_f$nf.err; }
run C2 2 --sort-mode visible_in_flight; run C2 3 --sort-mode visible_in_flight; run C2 4 --sort-mode visible_in_flight
run C5 2 --sort-mode both; run C5 2 --sort-mode visible_in_flight; run C5 4 --sort-mode visible_in_flight
run C3 2 --sort-mode visible_in_flight; run C4 2 --sort-mode visible_in_flight; run C2d 2 --sort-mode visible_in_flight
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_c4_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, {m:(x['ms_per_step'], x['regions_ms_per_step']) for m,x in d['modes'].items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-400:])
PY

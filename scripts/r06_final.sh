#!/bin/bash
# round 6, the final call on the final build: the whole -m gpu suite + smoke(), the measurement (scripts/r06_measure2.sh: the driver's command, the default run, C3,
# rocprofv3 stats + PMC passes of the driver's frames), and every mode at C5 / C2d (C2 / C3 / C4 on this build: scripts/r06_call23.sh)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06_pytest_gpu.log 2>&1; tail -3 gpurun_out/r06_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -1 gpurun_out/r06_smoke.log
bash scripts/r06_measure2.sh
for cfg in C5 C2d; do
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 --repeats 3 --sort-mode all --cpu-baseline off --pmc off > gpurun_out/r06_sf_$cfg.json 2> gpurun_out/r06_sf_$cfg.err
  python - $cfg <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/r06_sf_{sys.argv[1]}.json') if l.startswith('{')][-1])
    c = d["sort_mode_cross_check"] or {}
    print(sys.argv[1], {m: x["ms_per_step"] for m, x in d["modes"].items()}, 'cross', c.get("ok"), (c.get("in_flight") or {}).get("ok"))
except Exception as e:
    print(sys.argv[1], 'no line', e); print(open(f'gpurun_out/r06_sf_{sys.argv[1]}.err').read()[-1200:])
PY
done

#!/bin/bash
# round 6, call 21: every partition from one counter when a pass walks more partitions than it has workgroups (the hang of the in-flight mode at C3 / C5 / C2d / C4),
# lanes in GS_SORT_FULL (reference_shaped_in_flight): tests, then every configuration's bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_gpu_vissort.py tests/test_gpu_sort.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -6
for cfg in C2 C3 C5 C2d C4; do
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 --repeats 3 --sort-mode all --cpu-baseline off --pmc off > gpurun_out/r06_all_$cfg.json 2> gpurun_out/r06_all_$cfg.err
  python - $cfg <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/r06_all_{sys.argv[1]}.json') if l.startswith('{')][-1])
    c = d["sort_mode_cross_check"] or {}
    print(sys.argv[1], {m: x["ms_per_step"] for m, x in d["modes"].items()}, 'cross', c.get("ok"), (c.get("in_flight") or {}).get("ok"), (c.get("reference_shaped_in_flight") or {}).get("ok"))
except Exception as e:
    print(sys.argv[1], 'no line', e); print(open(f'gpurun_out/r06_all_{sys.argv[1]}.err').read()[-1200:])
PY
done

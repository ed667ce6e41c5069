#!/bin/bash
# round 6, call 23: the static first round back for a context that shares the GPU with nobody (bin_emit and the multi-round Onesweep passes; one counter stays the
# shared-GPU / lanes form): sort + visible-sort + full-size suites, every mode at C2 / C3 / C4, and the conversion kinds of the VALU issue probe
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_gpu_vissort.py tests/test_gpu_sort.py tests/test_gpu_fullsize.py tests/test_gpu_draw.py -x -q -m gpu 2>&1 | tail -4
for cfg in C2 C3 C4; do
  timeout 900 python bench.py --config $cfg --steps 20 --warmup 5 --repeats 3 --sort-mode all --cpu-baseline off --pmc off > gpurun_out/r06_sf_$cfg.json 2> gpurun_out/r06_sf_$cfg.err
  python - $cfg <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f'gpurun_out/r06_sf_{sys.argv[1]}.json') if l.startswith('{')][-1])
    c = d["sort_mode_cross_check"] or {}
    print(sys.argv[1], {m: x["ms_per_step"] for m, x in d["modes"].items()}, 'cross', c.get("ok"), (c.get("in_flight") or {}).get("ok"))
    print('   full stages', d["modes"]["full"].get("stages_ms") or d["modes"]["full"].get("stages"))
except Exception as e:
    print(sys.argv[1], 'no line', e); print(open(f'gpurun_out/r06_sf_{sys.argv[1]}.err').read()[-1200:])
PY
done
timeout 300 scripts/probes/valu_issue --from 6 > gpurun_out/r06_valu_issue_cvt.txt 2>&1; grep -E 'ind (4|8)' gpurun_out/r06_valu_issue_cvt.txt

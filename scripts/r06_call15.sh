#!/bin/bash
# round 6, call 15: issue priority of the latency-bound kernels against the blend's, with frames in flight (variants alternating on one box) + prefetch parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
GSPLAT_LIB_A=$PWD/unitygaussiansplatting_amd/libgsplat_hip.so GSPLAT_LIB_B=$PWD/unitygaussiansplatting_amd/variants/prefetch.so timeout 300 python scripts/ab_blend.py 2>&1 | head -5 | tee gpurun_out/r06_ab_prefetch_parity.txt
: > gpurun_out/r06_ab_prio.log
for rep in 1 2; do
for v in default cp3 cp3nb cp3v2 nb prefetch cp3pf; do
  L=$PWD/unitygaussiansplatting_amd/variants/$v.so; [ $v = default ] && L=$PWD/unitygaussiansplatting_amd/libgsplat_hip.so
  for cfg in C2 C5; do
    GSPLAT_LIB=$L timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 --repeats 3 --sort-mode all --cpu-baseline off --pmc off 2>/dev/null | grep '^{' > /tmp/line.json
    python - $v $cfg <<'PY' | tee -a gpurun_out/r06_ab_prio.log
import json, sys
d = json.loads(open('/tmp/line.json').read())
print(json.dumps({"lib": sys.argv[1], "cfg": sys.argv[2], **{m: x["ms_per_step"] for m, x in d["modes"].items()}, "ok": d["sort_mode_cross_check"]["ok"]}))
PY
  done
done; done

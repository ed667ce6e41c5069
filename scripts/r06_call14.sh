#!/bin/bash
# round 6, call 14: the blend's per-tile timeline (stats build) and the survivor-prefetch variant against the shipped kernel (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
GSPLAT_LIB=$PWD/unitygaussiansplatting_amd/variants/stats.so timeout 300 python scripts/blend_stats.py C2 1 visible 2>&1 | grep -v '^{' | tee gpurun_out/r06_blend_timeline_c2.txt
GSPLAT_LIB_A=$PWD/unitygaussiansplatting_amd/libgsplat_hip.so GSPLAT_LIB_B=$PWD/unitygaussiansplatting_amd/variants/prefetch.so timeout 300 python scripts/ab_blend.py 2>&1 | tail -12 | tee gpurun_out/r06_ab_prefetch_parity.txt
bash scripts/ab_variants.sh "C2 C3" 2 prefetch.so > /dev/null 2>&1
cp gpurun_out/ab_variants.log gpurun_out/r06_ab_prefetch.log
python - <<'PY'
import json
for l in open('gpurun_out/r06_ab_prefetch.log'):
    d = json.loads(l); print(d['cfg'], d['lib'], 'blend', d.get('blend'), 'wall', d.get('wall_min'), d.get('wall_med'))
PY

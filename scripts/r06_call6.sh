#!/bin/bash
# round 6, call 6: the blend's fp16 rounding as v_fma_mix_f32 + v_cvt_pk_f16_f32 (variants/cvtpk.so) against v_fma_mixlo/hi_f16 (shipped), same box, frame CRCs
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD; mkdir -p gpurun_out
for cfg in C2 C3 C2d; do for rep in 1 2; do
  python scripts/ab_sortmode.py $cfg 30 1 2>/dev/null | grep visible
  GSPLAT_LIB=$PWD/unitygaussiansplatting_amd/variants/cvtpk.so python scripts/ab_sortmode.py $cfg 30 1 2>/dev/null | grep visible
done; done > gpurun_out/r06_ab_cvtpk.log
python - <<'PY'
import json
for l in open('gpurun_out/r06_ab_cvtpk.log'):
    d=json.loads(l); print(d['cfg'], d['lib'], 'wall', d['wall_med'], 'blend', d['blend'], 'crc', d['frame_crc'])
PY

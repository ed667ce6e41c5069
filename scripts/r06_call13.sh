#!/bin/bash
# round 6, call 13: the automatic shared-GPU form (two contexts of one process in GS_SORT_FULL) + the blend's own work counters (-DGS_BLEND_STATS build)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sort.py -x -q -m gpu 2>&1 | tail -4
for c in C2 C3 C2d; do
  GSPLAT_LIB=$PWD/unitygaussiansplatting_amd/variants/stats.so timeout 300 python scripts/blend_stats.py $c 2 visible 2>&1 | grep '^{' | tee -a gpurun_out/r06_blend_stats.jsonl
done

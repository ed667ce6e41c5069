#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
for c in C2 C3; do GSPLAT_LIB=$PWD/unitygaussiansplatting_amd/variants/tl.so timeout 300 python scripts/blend_stats.py $c 1 visible 2>&1 | grep -v '^{' | tee gpurun_out/r06_blend_timeline_$c.txt; done

#!/bin/bash
# round 4, call 1: the tile-shape compositor -- parity tests, then the A/B of the three shapes (and the round-3 build) at C2 / C3 / C2d / C4
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_draw.py tests/test_golden.py tests/test_cutouts.py tests/test_scene_depth_and_debug.py tests/test_bc7.py -m gpu -q -x > $O/pytest_call1.log 2>&1; tail -5 $O/pytest_call1.log
: > $O/ab_tiles.log
for c in C2 C2d C3 C4; do
  fr=30; [ $c = C4 ] && fr=10
  timeout 600 python scripts/ab_tiles.py $c $fr 2>&1 | grep '^{' | tee -a $O/ab_tiles.log
  GSPLAT_LIB=$PWD/unitygaussiansplatting_amd/variants/r03.so timeout 400 python scripts/ab_tiles.py $c $fr 16x16 2>&1 | grep '^{' | tee -a $O/ab_tiles.log
done

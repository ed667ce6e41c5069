#!/bin/bash
# round 4, call 6: blend waves leave when their quadrant is finished (default) vs stay to the end of the tile (noexit)
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
timeout 180 python -m pytest tests/test_gpu_draw.py -m gpu -q -x -k "parity or shapes" > $O/pytest_call6a.log 2>&1; rc=$?; tail -4 $O/pytest_call6a.log
if [ $rc -ne 0 ]; then echo "draw tests failed or hung (rc $rc): stopping"; exit 1; fi
timeout 900 python -m pytest tests/test_gpu_draw.py tests/test_golden.py tests/test_cutouts.py tests/test_scene_depth_and_debug.py tests/test_bc7.py tests/test_gpu_configs.py -m gpu -q -x > $O/pytest_call6.log 2>&1; tail -4 $O/pytest_call6.log
: > $O/ab_call6.log
for rep in 1 2; do
for c in C2 C3 C4 C2d; do
  fr=30; [ $c = C4 ] && fr=10
  for v in default noexit; do
    L=""; [ $v != default ] && L=$V/$v.so
    GSPLAT_LIB=$L timeout 600 python scripts/ab_tiles.py $c $fr 16x16 32x16 2>&1 | grep '^{' | tee -a $O/ab_call6.log
  done
done; done

#!/bin/bash
# round 4, call 3: ref-parity debug, the rest of the -m gpu suite, tile-shape A/B of the sub-block-list blend
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
timeout 300 python scripts/debug_ref_gpu.py C1 > $O/debug_ref.log 2>&1; tail -25 $O/debug_ref.log
timeout 900 python -m pytest tests/test_gpu_draw.py tests/test_gpu_ref.py tests/test_gpu_sort.py tests/test_gpu_view.py tests/test_import.py tests/test_parallel.py tests/test_scene_depth_and_debug.py tests/test_validator.py -m gpu -q > $O/pytest_call3.log 2>&1; tail -8 $O/pytest_call3.log
: > $O/ab_tiles.log
for c in C2 C3 C2d C4; do
  fr=30; [ $c = C4 ] && fr=10
  timeout 600 python scripts/ab_tiles.py $c $fr 2>&1 | grep '^{' | tee -a $O/ab_tiles.log
done

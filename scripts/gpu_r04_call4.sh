#!/bin/bash
# round 4, call 4: static one-round tickets (E1), per-tile LDS counters in bin_emit (E5), XCD blocks in every sort pass, blend wave activity
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
timeout 600 python -m pytest tests/test_gpu_sort.py tests/test_gpu_draw.py tests/test_gpu_ref.py -m gpu -q -x > $O/pytest_call4.log 2>&1; tail -4 $O/pytest_call4.log
GSPLAT_LIB=$V/xcdall.so timeout 600 python -m pytest tests/test_gpu_sort.py -m gpu -q -x > $O/pytest_call4_xcd.log 2>&1; tail -2 $O/pytest_call4_xcd.log
: > $O/ab_call4.log
for rep in 1 2; do
for c in C2 C4; do
  fr=30; [ $c = C4 ] && fr=10
  for v in default noE1 noE5 xcdall; do
    L=""; [ $v != default ] && L=$V/$v.so
    GSPLAT_LIB=$L timeout 600 python scripts/ab_tiles.py $c $fr 32x16 2>&1 | grep '^{' | tee -a $O/ab_call4.log
  done
done; done
for t in 16x16 32x16; do
  TILE=$t GSPLAT_LIB=$V/tl.so timeout 300 python scripts/blend_timeline.py C2 > $O/blend_tl_$t.txt 2>&1; tail -25 $O/blend_tl_$t.txt
done

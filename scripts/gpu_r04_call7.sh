#!/bin/bash
# round 4, call 7: bin_emit -- one request per poll while waiting for the group's base (default) vs the 64-lane poll (noquiet); 12 positions per thread
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
timeout 300 python -m pytest tests/test_gpu_draw.py -m gpu -q -x > $O/pytest_call7.log 2>&1; tail -3 $O/pytest_call7.log
: > $O/ab_call7.log
for rep in 1 2; do
for c in C2 C4 C2d; do
  fr=30; [ $c = C4 ] && fr=10
  for v in default noquiet items12; do
    L=""; [ $v != default ] && L=$V/$v.so
    GSPLAT_LIB=$L timeout 600 python scripts/ab_tiles.py $c $fr auto 2>&1 | grep '^{' | tee -a $O/ab_call7.log
  done
done; done
GSPLAT_LIB=$V/bintl.so timeout 300 python scripts/bin_timeline.py C2 > $O/bintl_quiet.log 2>&1; head -12 $O/bintl_quiet.log
GSPLAT_LIB=$V/bintl_nq.so timeout 300 python scripts/bin_timeline.py C2 > $O/bintl_noquiet.log 2>&1; head -12 $O/bintl_noquiet.log

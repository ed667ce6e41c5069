#!/bin/bash
# round 4, call 11: one-round look-back with eight group aggregates per round (default) vs four + inclusive prefixes (prevlb)
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
timeout 600 python -m pytest tests/test_gpu_sort.py -m gpu -q -x > $O/pytest_call11.log 2>&1; tail -3 $O/pytest_call11.log
: > $O/ab_call11.log
for rep in 1 2 3; do
for c in C2 C3; do
  for v in default prevlb; do
    L=""; [ $v != default ] && L=$V/$v.so
    GSPLAT_LIB=$L timeout 600 python scripts/ab_tiles.py $c 30 auto 2>&1 | grep '^{' | tee -a $O/ab_call11.log
  done
done; done

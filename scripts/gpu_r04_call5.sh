#!/bin/bash
# round 4, call 5: team look-back in the one-round regime (default: 4 aggregates per round; team8: 8; noteam: the previous look-back)
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
timeout 600 python -m pytest tests/test_gpu_sort.py tests/test_gpu_draw.py -m gpu -q -x > $O/pytest_call5.log 2>&1; tail -4 $O/pytest_call5.log
GSPLAT_LIB=$V/team8.so timeout 600 python -m pytest tests/test_gpu_sort.py -m gpu -q -x > $O/pytest_call5_t8.log 2>&1; tail -2 $O/pytest_call5_t8.log
: > $O/ab_call5.log
for rep in 1 2 3; do
for c in C2 C3; do
  for v in default noteam team8; do
    L=""; [ $v != default ] && L=$V/$v.so
    GSPLAT_LIB=$L timeout 600 python scripts/ab_tiles.py $c 30 32x16 2>&1 | grep '^{' | tee -a $O/ab_call5.log
  done
done; done

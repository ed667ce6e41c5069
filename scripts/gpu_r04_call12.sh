#!/bin/bash
# round 4, call 12: sort_keys_kernel -- ballot-aggregated LDS histogram adds for key bytes that take few values inside a wave (default: byte 2; few12: bytes 2, 3; few14: 1, 2, 3; few0: none)
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
timeout 600 python -m pytest tests/test_gpu_sort.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest_call12.log 2>&1; tail -3 $O/pytest_call12.log
: > $O/ab_call12.log
for rep in 1 2; do
for c in C2 C3 C4; do
  fr=30; [ $c = C4 ] && fr=10
  for v in default few0 few12 few14; do
    L=""; [ $v != default ] && L=$V/$v.so
    GSPLAT_LIB=$L timeout 600 python scripts/ab_tiles.py $c $fr auto 2>&1 | grep '^{' | tee -a $O/ab_call12.log
  done
done; done

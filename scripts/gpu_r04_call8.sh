#!/bin/bash
# round 4, call 8: bin_emit positions per thread (8 default / 12 / 16) and resident workgroups per CU
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
: > $O/ab_call8.log
for rep in 1 2; do
for c in C2 C4 C2d C3; do
  fr=30; [ $c = C4 ] && fr=10
  for v in default items12 i12b4 i16b4 i12w5; do
    L=""; [ $v != default ] && L=$V/$v.so
    GSPLAT_LIB=$L timeout 600 python scripts/ab_tiles.py $c $fr auto 2>&1 | grep '^{' | tee -a $O/ab_call8.log
  done
done; done

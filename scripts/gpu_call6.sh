#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
GSPLAT_LIB=$V/oneshot.so timeout 300 python scripts/gpu_quickcheck.py 200000 1280 720 2>&1 | grep -E "order|mode|view" | tee $O/quick_oneshot.log
: > $O/variants.log
for cfgk in C2 C3; do
  timeout 300 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  GSPLAT_LIB=$V/oneshot.so timeout 300 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
done
GSPLAT_LIB=$V/bintl.so timeout 300 python scripts/bin_timeline.py C2 > $O/bintl_default.log 2>&1; tail -22 $O/bintl_default.log
GSPLAT_LIB=$V/oneshottl.so timeout 300 python scripts/bin_timeline.py C2 > $O/bintl_oneshot.log 2>&1; tail -22 $O/bintl_oneshot.log

#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
for v in oneshot pipe; do GSPLAT_LIB=$V/$v.so timeout 300 python scripts/gpu_quickcheck.py 200000 1280 720 2>&1 | grep -E "mode" | tee -a $O/quick7.log; done
: > $O/variants.log
for cfgk in C2 C3; do
  timeout 300 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  for v in oneshot pipe; do GSPLAT_LIB=$V/$v.so timeout 300 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log; done
done
GSPLAT_LIB=$V/pipetl.so timeout 300 python scripts/bin_timeline.py C2 > $O/bintl_pipe.log 2>&1; tail -22 $O/bintl_pipe.log

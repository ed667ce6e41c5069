#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
timeout 300 python scripts/gpu_quickcheck.py 200000 1280 720 2>&1 | grep -E "mode" | tee $O/quick9.log
: > $O/variants.log
for rep in 1 2; do for cfgk in C2 C3; do
  timeout 300 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  GSPLAT_LIB=$V/prev.so timeout 300 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
done; done

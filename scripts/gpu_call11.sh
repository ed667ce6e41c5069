#!/bin/bash
# A/B of the in-tree build against variants/prev.so: stage times per configuration ($@ = configs, default C2 C3)
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
: > $O/variants.log
for rep in 1 2; do for cfgk in ${@:-C2 C3}; do
  timeout 400 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  GSPLAT_LIB=$V/prev.so timeout 400 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
done; done

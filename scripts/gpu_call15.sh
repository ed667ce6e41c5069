#!/bin/bash
# stage times of the in-tree build and of the variants in $VARS ($@ = configs), two repetitions
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
: > $O/variants.log
for rep in 1 2; do for cfgk in ${@:-C2}; do
  timeout 400 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  for v in $VARS; do
    GSPLAT_LIB=$V/$v.so timeout 400 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  done
done; done

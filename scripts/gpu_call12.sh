#!/bin/bash
# A/B of variant libraries (names in $VARS) against the in-tree build: stage times per configuration ($@ = configs)
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
: > $O/variants.log
for v in $VARS; do
  GSPLAT_LIB=$V/$v.so timeout 600 python -m pytest tests/test_gpu_sort.py -m gpu -q -x 2>&1 | tail -1 | tee -a $O/variants.log
done
for rep in 1 2; do for cfgk in ${@:-C2}; do
  timeout 400 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  for v in $VARS; do
    GSPLAT_LIB=$V/$v.so timeout 400 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  done
done; done

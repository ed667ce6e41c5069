#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
GSPLAT_SKIP_C4=1 timeout 1200 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -9 $O/pytest_gpu.log
for v in pipe4 loop2; do GSPLAT_LIB=$PWD/unitygaussiansplatting_amd/variants/$v.so timeout 300 python scripts/gpu_quickcheck.py 200000 1280 720 2>&1 | grep -E "order|mode|view" | tee -a $O/quick_variants.log; done
: > $O/variants.log
for cfgk in C2 C3; do
  timeout 300 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  for v in unitygaussiansplatting_amd/variants/*.so; do
    GSPLAT_LIB=$PWD/$v timeout 300 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  done
done

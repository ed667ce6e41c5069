#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
GSPLAT_SKIP_C4=1 timeout 900 python -m pytest tests/test_gpu_view.py tests/test_gpu_configs.py tests/test_gpu_draw.py tests/test_cutouts.py tests/test_bc7.py -m gpu -q -x > $O/pytest_view.log 2>&1; tail -3 $O/pytest_view.log
: > $O/variants.log
for rep in 1 2; do for cfgk in C2 C3; do
  timeout 300 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  GSPLAT_LIB=$V/prev.so timeout 300 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
done; done

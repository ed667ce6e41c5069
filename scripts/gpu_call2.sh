#!/bin/bash
# round-3 call: full parity tests, quick parity of the `groups` blend variant, C2 / C3 stage A/B of the variants
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
GSPLAT_LIB=$PWD/unitygaussiansplatting_amd/variants/groups.so timeout 300 python scripts/gpu_quickcheck.py 200000 1280 720 2>&1 | tail -6 | tee $O/quick_groups.log
: > $O/variants.log
for cfgk in C2 C3; do
  timeout 300 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  for v in unitygaussiansplatting_amd/variants/*.so; do
    GSPLAT_LIB=$PWD/$v timeout 300 python scripts/bench_stages.py $cfgk 30 2>&1 | tail -1 | tee -a $O/variants.log
  done
done

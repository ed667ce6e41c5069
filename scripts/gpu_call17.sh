#!/bin/bash
# the two-shape Onesweep: sort tests in every shape, the configuration tests (C4 takes shape B by itself), then C4 / C2 stage times vs prev.so
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sort.py tests/test_gpu_configs.py -m gpu -q -x > $O/pytest_sort.log 2>&1; tail -3 $O/pytest_sort.log
VARS="prev" bash scripts/gpu_call15.sh C4 C2 2>&1 | grep -v "^$"

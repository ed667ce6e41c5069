#!/bin/bash
# full GPU suite on the in-tree build, then A/B against variants/prev.so ($@ = configs)
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
bash scripts/gpu_call11.sh "$@"

#!/bin/bash
# round 6, call 33: what is left of the GPU budget: 85 further seeds of the random parity walk (2000 ...)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
GSPLAT_PARITY_SEED0=2000 GSPLAT_PARITY_SEEDS=85 timeout 175 python -m pytest tests/test_gpu_random_parity.py -q -m gpu > gpurun_out/r06_random_parity_3.log 2>&1; tail -3 gpurun_out/r06_random_parity_3.log | cut -c1-400
grep -E '^FAILED|^E  ' gpurun_out/r06_random_parity_3.log | cut -c1-400 | head

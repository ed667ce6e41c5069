#!/bin/bash
# round 6, call 31: the random parity walk, 230 further seeds (1000 ...)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
GSPLAT_PARITY_SEED0=1000 GSPLAT_PARITY_SEEDS=230 timeout 460 python -m pytest tests/test_gpu_random_parity.py -q -m gpu > gpurun_out/r06_random_parity_2.log 2>&1; tail -5 gpurun_out/r06_random_parity_2.log | cut -c1-600
grep -E '^FAILED|^E  ' gpurun_out/r06_random_parity_2.log | cut -c1-500 | head -20

#!/bin/bash
# round 6, call 27: the random parity walk (tests/test_gpu_random_parity.py): the suite's three seeds + a campaign of 150 more
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
GSPLAT_PARITY_SEEDS=150 timeout 720 python -m pytest tests/test_gpu_random_parity.py -v -q -m gpu > gpurun_out/r06_random_parity.log 2>&1; tail -40 gpurun_out/r06_random_parity.log | cut -c1-700
grep -c PASSED gpurun_out/r06_random_parity.log; grep -E '^FAILED|passed|failed' gpurun_out/r06_random_parity.log | cut -c1-300 | tail -40

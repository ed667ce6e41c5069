#!/bin/bash
# round 6, call 24: random API sequences through the lanes (tests/test_gpu_lanes_fuzz.py): the suite's three seeds + a campaign of 24 more
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
GSPLAT_FUZZ_SEEDS=24 timeout 1500 python -m pytest tests/test_gpu_lanes_fuzz.py -q -m gpu > gpurun_out/r06_lanes_fuzz.log 2>&1; tail -60 gpurun_out/r06_lanes_fuzz.log | cut -c1-400

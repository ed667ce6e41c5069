#!/bin/bash
# the -m gpu suite's slowest tests (what a time limit on the driver's side would meet first)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
timeout 2400 python -m pytest tests -x -q -m gpu --durations=60 > gpurun_out/r06_pytest_gpu_durations.log 2>&1; tail -75 gpurun_out/r06_pytest_gpu_durations.log

#!/bin/bash
# round 6, call 28: the whole -m gpu suite on the final build with the host layer's default sort mode switched to the visible-only path
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
GSPLAT_SORT_MODE=visible timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06_pytest_gpu_visible.log 2>&1; tail -4 gpurun_out/r06_pytest_gpu_visible.log

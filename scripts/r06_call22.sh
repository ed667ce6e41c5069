#!/bin/bash
# round 6, call 22: the final build (lanes in GS_SORT_VISIBLE only, every partition from one counter in multi-round passes): the whole -m gpu suite, smoke(), the suite
# again with the host layer's default sort mode switched to the visible-only path
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06_pytest_gpu.log 2>&1; tail -4 gpurun_out/r06_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -1 gpurun_out/r06_smoke.log
GSPLAT_SORT_MODE=visible timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06_pytest_gpu_visible.log 2>&1; tail -4 gpurun_out/r06_pytest_gpu_visible.log

#!/bin/bash
# round 6, call 5: the whole -m gpu suite (once as is, once with the host layer's default sort mode = visible), smoke
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -30 > gpurun_out/r06_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1
GSPLAT_SORT_MODE=visible timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r06_pytest_gpu_visible.log
tail -4 gpurun_out/r06_pytest_gpu.log; tail -2 gpurun_out/r06_smoke.log; tail -4 gpurun_out/r06_pytest_gpu_visible.log

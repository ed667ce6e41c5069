#!/bin/bash
# round 6, call 19: the whole -m gpu suite + smoke() on the build with the lanes inside the library (ABI 9)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$PWD
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06_pytest_gpu.log 2>&1; tail -6 gpurun_out/r06_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -2 gpurun_out/r06_smoke.log

#!/bin/bash
# round 6, call 1: the visible-sort suite on the 9-bit passes, the sort suite, the probe table, the driver's bench command
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vissort.py tests/test_gpu_sort.py -x -q -m gpu --durations=12 2>&1 | tail -30 > gpurun_out/r06_c1_pytest.log
./scripts/probes/valu_issue > gpurun_out/r06_valu_issue.txt 2>&1
./scripts/probes/valu_issue --quick > gpurun_out/r06_valu_issue_quick.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_c1_bench.json 2> gpurun_out/r06_c1_bench.err
tail -5 gpurun_out/r06_c1_pytest.log; tail -c 1500 gpurun_out/r06_c1_bench.err; wc -c gpurun_out/r06_c1_bench.json

#!/bin/bash
# round 4, call 2: the whole -m gpu suite on the tile-shape build, then the new bench.py line (C2, no CPU baseline)
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --cpu-baseline off > $O/bench_quick.json 2> $O/bench_quick.err; tail -c 3000 $O/bench_quick.json; tail -3 $O/bench_quick.err

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_inflight; rm -rf $O; mkdir -p $O; export PYTHONPATH=$R
rocprofv3 --kernel-trace --output-format csv -d $O -- python $R/bench.py --steps 20 --warmup 5 --sort-mode visible_in_flight --cpu-baseline off --pmc off --repeats 1 > $O/bench.json 2> $O/bench.err
F=$(find $O -name "*kernel_trace.csv" | head -1); head -1 $F | cut -c1-400
python $R/scripts/inflight_trace.py $F | tee $R/gpurun_out/r06_inflight_trace.txt

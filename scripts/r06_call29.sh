#!/bin/bash
# round 6, call 29: kernel trace of the headline mode on the final build (one renderer, two lanes inside the library): what runs beside what
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_inflight_library; rm -rf $O; mkdir -p $O; export PYTHONPATH=$R
rocprofv3 --kernel-trace --output-format csv -d $O -- python $R/bench.py --steps 20 --warmup 5 --sort-mode all --cpu-baseline off --pmc off --repeats 1 > $O/bench.json 2> $O/bench.err
F=$(find $O -name "*kernel_trace.csv" | head -1); head -1 $F | cut -c1-300
python $R/scripts/inflight_trace.py $F 0.85 | tee $R/gpurun_out/r06_inflight_trace_library.txt
tail -2 $O/bench.err

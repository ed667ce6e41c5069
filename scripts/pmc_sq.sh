#!/bin/bash
# SQ wave-cycle breakdown per kernel (issue-bound vs latency-bound).  Outputs under gpurun_out/pmc_sq/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
rocprofv3 -L > $O/counters.txt 2>&1
CMD="python $R/scripts/bench_stages.py ${1:-C2} ${2:-55}"      # frames 5 .. 59: ends on the default bench.py run's last frame (same pair count)
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/p2 -- $CMD > $O/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/p3 -- $CMD > $O/p3.log 2>&1
tail -3 $O/p1.log $O/p2.log $O/p3.log
find $O -name "*counter_collection.csv" | head

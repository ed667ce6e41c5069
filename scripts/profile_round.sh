#!/bin/bash
# (remove the local gpurun_out/prof first: gpurun MERGES what the box wrote into it)
# On the GPU box: kernel-trace stats of the default bench, then separate PMC passes (FETCH_SIZE, WRITE_SIZE) for the bench
# and for the calibration probe, then one SQ_INSTS_VALU pass.  Usage: profile_round.sh [C2|C4|...] [steps] [visible|full] [warmup] [key suffix].
# Outputs under gpurun_out/prof_<config>[_visible][suffix]/.  The driver times `bench.py --gpus 1 --steps 20 --warmup 5` (other frames than the default
# run, hence another pair count): `profile_round.sh C2 20 visible 5 @s20w5` collects the counters bench.py pairs with THAT line.
cd /tmp && export TMPDIR=/tmp
CFG=${1:-C2}; STEPS=${2:-50}; MODE=${3:-visible}; WARM=${4:-10}; KSUF=${5:-}      # 50 steps after 10 warm-up frames = the default bench.py run: the same frames, the same pair count
SUF=""; [ $MODE = visible ] && SUF=_visible
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${CFG}${SUF}${KSUF}; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
BENCH="python $R/bench.py --config $CFG --steps $STEPS --warmup $WARM --cpu-baseline off --pmc off --sort-mode $MODE --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/bench_stats.json 2> $O/bench_stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- $BENCH > $O/bench_$c.json 2> $O/bench_$c.err
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$c -- $R/scripts/probes/pmc_calib > $O/calib_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_SQ_INSTS_VALU -- $BENCH > $O/bench_SQ_INSTS_VALU.json 2> $O/bench_SQ_INSTS_VALU.err
find $O -name "*.csv" | head -40

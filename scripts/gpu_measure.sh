#!/bin/bash
# The measurement call of a round (on the GPU box, from the repo root): bench lines of every single-GPU configuration,
# rocprofv3 kernel stats + PMC traffic passes of the headline bench, SQ counters.  Outputs under gpurun_out/.
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
WHAT="${@:-c2 c3 c5 c2d prof sq c4 prof4}"
for w in $WHAT; do
case $w in
c2) timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 600 $O/bench_c2.json;;
c3) timeout 600 python bench.py --config C3 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 300 $O/bench_c3.json;;
c4) timeout 900 python bench.py --config C4 --steps 20 > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 300 $O/bench_c4.json;;
c5) timeout 600 python bench.py --config C5 --steps 20 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 300 $O/bench_c5.json;;
c2d) timeout 600 python bench.py --config C2d --steps 20 > $O/bench_c2d.json 2> $O/bench_c2d.err; tail -c 300 $O/bench_c2d.json;;
prof) bash scripts/profile_round.sh C2 50 visible > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log;;
proff) bash scripts/profile_round.sh C2 50 full > $O/profile_round_full.log 2>&1; tail -5 $O/profile_round_full.log;;
prof4) bash scripts/profile_round.sh C4 6 visible > $O/profile_round_c4.log 2>&1; tail -5 $O/profile_round_c4.log;;
sq) bash scripts/pmc_sq.sh C2 > $O/pmc_sq.log 2>&1; tail -3 $O/pmc_sq.log;;
esac
done

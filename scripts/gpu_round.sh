#!/bin/bash
# One GPU-box call: parity tests, A/B stage timings of the library variants, rocprofv3 kernel stats, the bench line.
# Usage (from the repo root on the box): bash scripts/gpu_round.sh [tests] [tests_noc4] [variants] [prof] [bench] [stages]
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
WHAT="${@:-tests variants prof bench}"
for w in $WHAT; do
case $w in
tests)
  timeout 1700 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log;;
tests_noc4)
  GSPLAT_SKIP_C4=1 timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log;;
stages)
  timeout 300 python scripts/bench_stages.py C2 30 2>&1 | tail -1 | tee $O/stages.log;;
variants)
  : > $O/variants.log
  timeout 300 python scripts/bench_stages.py C2 30 2>&1 | tail -1 | tee -a $O/variants.log
  for v in unitygaussiansplatting_amd/variants/*.so; do
    [ -e "$v" ] || continue
    GSPLAT_LIB=$PWD/$v timeout 200 python scripts/bench_stages.py C2 30 2>&1 | tail -1 | tee -a $O/variants.log
  done;;
overlap)
  : > $O/overlap.log
  for cfgk in C2 C3; do for e in "GSPLAT_OVERLAP=0" "GSPLAT_OVERLAP=1" "GSPLAT_OVERLAP=0" "GSPLAT_OVERLAP=1"; do
    echo -n "$cfgk $e  " | tee -a $O/overlap.log
    env $e GS_NOPROF=1 timeout 300 python scripts/bench_stages.py $cfgk 100 2>&1 | tail -1 | tee -a $O/overlap.log
  done; done;;
prof)
  (cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof && mkdir -p $O/prof && R=$GRAFT_REPO_ROOT && \
   timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/stats -- python $R/bench.py --steps 20 --warmup 5 --cpu-baseline off > $O/prof/bench_stats.json 2> $O/prof/bench_stats.err)
  find $O/prof -name "*kernel_stats.csv" | head; tail -c 600 $O/prof/bench_stats.json;;
bench)
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json;;
esac
done

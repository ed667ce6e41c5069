#!/bin/bash
# On the GPU box (one gpurun call = one box): HEAD's library against variant builds, alternating, same frames.
#   bash scripts/ab_variants.sh "C2 C3 C2d" [reps] variant1.so variant2.so ...     (variants: scripts/build_variants.py, unitygaussiansplatting_amd/variants/)
# One JSON line per (configuration, build, repetition) -> gpurun_out/ab_variants.log: hipEvent stage means + the un-instrumented wall time per frame.
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
CFGS=$1; REPS=$2; shift 2
: > $O/ab_variants.log
for rep in $(seq 1 $REPS); do
for c in $CFGS; do
  fr=30; [ $c = C4 ] && fr=10
  timeout 600 python scripts/ab_tiles.py $c $fr auto 2>&1 | grep '^{' | tee -a $O/ab_variants.log
  for v in "$@"; do
    GSPLAT_LIB=$PWD/unitygaussiansplatting_amd/variants/$v timeout 600 python scripts/ab_tiles.py $c $fr auto 2>&1 | grep '^{' | tee -a $O/ab_variants.log
  done
done; done

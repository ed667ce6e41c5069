#!/bin/bash
# On the GPU box (one gpurun call = one box): the PREVIOUS round's build against HEAD, same process layout, same frames.
#   bash scripts/ab_rounds.sh [prev.so] [configs...]      default: unitygaussiansplatting_amd/variants/r04.so, C2 C3 C4 C2d
# Per configuration: the previous round's library (its only sort mode: all N splats), then HEAD in GS_SORT_FULL and GS_SORT_VISIBLE (one process,
# alternating: scripts/ab_sortmode.py) -- hipEvent stage means over 30 frames (10 for C4) and the un-instrumented wall time per frame (min / median of
# three regions), two repetitions, one JSON line each -> gpurun_out/ab_rounds.log (committed as profiles/rNN_ab_rounds.log).
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
PREV=${1:-$PWD/unitygaussiansplatting_amd/variants/r04.so}; shift
CFGS=${@:-C2 C3 C4 C2d}
: > $O/ab_rounds.log
for rep in 1 2; do
for c in $CFGS; do
  fr=30; [ $c = C4 ] && fr=10
  GSPLAT_LIB=$PREV timeout 600 python scripts/ab_tiles.py $c $fr auto 2>&1 | grep '^{' | tee -a $O/ab_rounds.log
  timeout 600 python scripts/ab_sortmode.py $c $fr 1 2>&1 | grep '^{' | tee -a $O/ab_rounds.log
done; done

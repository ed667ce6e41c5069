#!/bin/bash
# Review item 4 (round 5): the split-heaviest-tiles experiment against the shipped build, one call, C2 first.  Output: gpurun_out/ab_blend_split.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=$PWD/unitygaussiansplatting_amd/variants
out=gpurun_out/ab_blend_split.log
: > $out
for cfg in ${1:-C2 C2d C3}; do
    python scripts/ab_blend_split.py $cfg 30 0,3 2 >> $out 2>&1
    GSPLAT_LIB=$V/r05_split4.so  python scripts/ab_blend_split.py $cfg 30 0,1.5,3,8,16 2 >> $out 2>&1
    GSPLAT_LIB=$V/r05_split16.so python scripts/ab_blend_split.py $cfg 30 0,1.5,3,8,16 2 >> $out 2>&1
done
cat $out

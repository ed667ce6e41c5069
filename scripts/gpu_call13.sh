#!/bin/bash
# onesweep experiments: phase timelines of the variants in $TLS, then sort tests + stage times of the variants in $VARS
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
for t in $TLS; do
  GSPLAT_LIB=$V/$t.so timeout 200 python scripts/sort_timeline.py 6131954 8 2>&1 | head -14 > $O/sort_$t.txt; cat $O/sort_$t.txt
done
VARS="$VARS" bash scripts/gpu_call12.sh "$@"

#!/bin/bash
# On the GPU box: stage times of every library variant under unitygaussiansplatting_amd/variants (plus the default build).
export PYTHONPATH=$PWD
CFG=${1:-C2}; FR=${2:-30}
python scripts/bench_stages.py $CFG $FR 2>&1 | tail -1
for v in unitygaussiansplatting_amd/variants/*.so; do
  GSPLAT_LIB=$PWD/$v timeout 120 python scripts/bench_stages.py $CFG $FR 2>&1 | tail -1
done

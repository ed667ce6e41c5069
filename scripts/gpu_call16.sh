#!/bin/bash
# correctness of the variants in $VARS on the draw tests, then stage times ($@ = configs)
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
for v in $VARS; do
  GSPLAT_SKIP_C4=1 GSPLAT_LIB=$V/$v.so timeout 900 python -m pytest tests/test_gpu_draw.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -1
done
bash scripts/gpu_call15.sh "$@"

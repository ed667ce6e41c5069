#!/bin/bash
# round 4, call 10: the binning as count / scan / emit launches (default) vs the fused persistent kernel (fused)
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
timeout 240 python -m pytest tests/test_gpu_draw.py -m gpu -q -x > $O/pytest_call10a.log 2>&1; rc=$?; tail -4 $O/pytest_call10a.log
if [ $rc -ne 0 ]; then echo "draw tests failed or hung (rc $rc): stopping"; exit 1; fi
timeout 900 python -m pytest tests/test_golden.py tests/test_cutouts.py tests/test_scene_depth_and_debug.py tests/test_bc7.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest_call10.log 2>&1; tail -4 $O/pytest_call10.log
: > $O/ab_call10.log
for rep in 1 2; do
for c in C2 C3 C4 C2d; do
  fr=30; [ $c = C4 ] && fr=10
  for v in default fused; do
    L=""; [ $v != default ] && L=$V/$v.so
    GSPLAT_LIB=$L timeout 600 python scripts/ab_tiles.py $c $fr auto 2>&1 | grep '^{' | tee -a $O/ab_call10.log
  done
done; done
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_split -- python $GRAFT_REPO_ROOT/scripts/ab_tiles.py C2 20 auto > $O/prof_split.log 2>&1
find $O/prof_split -name "*kernel_stats*" | head -2 | while read f; do head -14 "$f" | cut -c1-160; done

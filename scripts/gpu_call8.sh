#!/bin/bash
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
V=$PWD/unitygaussiansplatting_amd/variants
GSPLAT_LIB=$V/split.so timeout 300 python scripts/gpu_quickcheck.py 200000 1280 720 2>&1 | grep -E "mode" | tee $O/quick8.log
: > $O/variants.log
timeout 300 python scripts/bench_stages.py C2 30 2>&1 | tail -1 | tee -a $O/variants.log
GSPLAT_LIB=$V/split.so timeout 300 python scripts/bench_stages.py C2 30 2>&1 | tail -1 | tee -a $O/variants.log
(cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof_split && GSPLAT_LIB=$V/split.so GS_NOPROF=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_split -- python $GRAFT_REPO_ROOT/scripts/bench_stages.py C2 30 > $O/prof_split.log 2>&1)
f=$(find $O/prof_split -name "*kernel_stats.csv" | head -1); cut -d, -f1-5 $f | head -20 | tee $O/split_kernel_stats.txt
GSPLAT_LIB=$V/splittl.so timeout 300 python scripts/bin_timeline.py C2 > $O/bintl_split.log 2>&1; tail -20 $O/bintl_split.log

#!/bin/bash
# round 6, call 7: framebuffer parity vs the oracle of the shipped blend (v_fma_mixlo/hi_f16) and of the cvt_pk variant
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD; mkdir -p gpurun_out
python scripts/gpu_parity_report.py > gpurun_out/r06_parity_default.jsonl 2>/dev/null
GSPLAT_LIB=$PWD/unitygaussiansplatting_amd/variants/cvtpk.so python scripts/gpu_parity_report.py > gpurun_out/r06_parity_cvtpk.jsonl 2>/dev/null
paste -d'\n' gpurun_out/r06_parity_default.jsonl gpurun_out/r06_parity_cvtpk.jsonl | cut -c1-260

export PYTHONPATH=$PWD
mkdir -p gpurun_out
python scripts/bench_stages.py C2 100 2>&1 | tail -1 | cut -c1-400
GS_NOPROF=1 python scripts/bench_stages.py C2 100 2>&1 | tail -1
GSPLAT_OVERLAP=1 python scripts/bench_stages.py C2 100 2>&1 | tail -1 | cut -c1-400
GSPLAT_OVERLAP=1 GS_NOPROF=1 python scripts/bench_stages.py C2 100 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_draw.py -x -q -m gpu -k overlap 2>&1 | tail -2

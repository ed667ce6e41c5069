export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python scripts/bench_stages.py C2 100 2>&1 | tail -1 | cut -c1-330
GS_NOPROF=1 python scripts/bench_stages.py C2 100 2>&1 | tail -1
for v in unitygaussiansplatting_amd/variants/*.so; do GSPLAT_LIB=$PWD/$v python scripts/bench_stages.py C2 100 2>&1 | tail -1 | cut -c1-60; done

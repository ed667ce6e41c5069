export PYTHONPATH=$PWD
mkdir -p gpurun_out
python scripts/bench_stages.py C2 100 2>&1 | tail -1 | cut -c1-330
GS_NOPROF=1 python scripts/bench_stages.py C2 100 2>&1 | tail -1
for v in unitygaussiansplatting_amd/variants/*.so; do GSPLAT_LIB=$PWD/$v python scripts/bench_stages.py C2 100 2>&1 | tail -1 | cut -c1-330; done
timeout 600 python -m pytest tests/test_gpu_draw.py tests/test_gpu_view.py -x -q -m gpu 2>&1 | tail -2

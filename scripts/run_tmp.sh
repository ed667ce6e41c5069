export PYTHONPATH=$PWD
mkdir -p gpurun_out
python scripts/bench_stages.py C2 100 2>&1 | tail -1 | cut -c1-330
for v in unitygaussiansplatting_amd/variants/*.so; do GSPLAT_LIB=$PWD/$v python scripts/bench_stages.py C2 100 2>&1 | tail -1 | cut -c1-330; done

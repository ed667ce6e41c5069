export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_draw.py tests/test_golden.py tests/test_gpu_fullsize.py tests/test_cutouts.py -x -q -m gpu 2>&1 | tail -3
python scripts/bench_stages.py C2 100 2>&1 | tail -1 | cut -c1-330
GS_NOPROF=1 python scripts/bench_stages.py C2 100 2>&1 | tail -1

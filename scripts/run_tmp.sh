export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_draw.py tests/test_golden.py tests/test_cutouts.py -x -q -m gpu 2>&1 | tail -3
GSPLAT_LIB=$PWD/unitygaussiansplatting_amd/variants/btl.so python scripts/blend_timeline.py C2 2>&1 | head -8
bash scripts/gpu_round.sh variants 2>&1 | grep -v "btl.so" | cut -c1-400

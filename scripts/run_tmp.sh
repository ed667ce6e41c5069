export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash scripts/gpu_round.sh variants 2>&1 | grep -v "timeline.so"

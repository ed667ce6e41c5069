export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
bash scripts/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -2 gpurun_out/profile_round.log
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 600 gpurun_out/bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

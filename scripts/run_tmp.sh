export PYTHONPATH=$PWD
mkdir -p gpurun_out
bash scripts/profile_round.sh > gpurun_out/profile_round.log 2>&1; tail -3 gpurun_out/profile_round.log
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 2500 gpurun_out/bench.json

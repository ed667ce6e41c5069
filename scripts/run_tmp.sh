export PYTHONPATH=$PWD
mkdir -p gpurun_out
free -g | head -2
(time timeout 600 python bench.py --config C3 --steps 30 --warmup 5 --cpu-baseline off) > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; tail -c 1500 gpurun_out/bench_c3.json; tail -3 gpurun_out/bench_c3.err
(time timeout 900 python bench.py --config C4 --steps 20 --warmup 5 --cpu-baseline off) > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; tail -c 1500 gpurun_out/bench_c4.json; tail -3 gpurun_out/bench_c4.err

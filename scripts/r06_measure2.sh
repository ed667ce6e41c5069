#!/bin/bash
# round 6, the measurement call on the final build (visible_in_flight = one renderer, lanes inside the library): the driver's command, the default run, every single-GPU
# configuration, rocprofv3 stats + PMC passes of the driver's frames
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD; O=$PWD/gpurun_out; mkdir -p $O
T0=$(date +%s.%N)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_cmd.json 2> $O/r06_bench_driver_cmd.err
T1=$(date +%s.%N); python -c "print(\"driver command wall: %.1f s\" % ($T1 - $T0))"
timeout 600 python bench.py > $O/r06_bench_c2.json 2> $O/r06_bench_c2.err
timeout 900 python bench.py --config C3 --steps 20 > $O/r06_bench_c3.json 2> $O/r06_bench_c3.err
# (C5 / C2d / C4 on this build: gpurun_out/r06_all_*.json of call 21, every mode back to back, no oracle replay, no counters)
bash scripts/profile_round.sh C2 20 visible 5 @s20w5 > $O/r06_profile_round.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['config']['sort_mode'], d['config'].get('frames_in_flight_impl'), {m:x['ms_per_step'] for m,x in d['modes'].items()}, (d.get('end_of_orbit_check') or {}).get('ok'), ((d.get('parity_vs_oracle') or {}).get('visible_in_flight') or {}).get('ok'), d['roofline']['frac'], d['roofline']['traffic'], 'P', d['modes']['visible']['tile_pairs_P'], 'V', d['modes']['visible']['visible_splats'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 $O/r06_profile_round.log

#!/bin/bash
# the driver's command, timed by the shell; prints the headline fields
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T0=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_cmd.json 2> gpurun_out/r06_bench_driver_cmd.err
T1=$(date +%s.%N); echo "bench wall: $(echo "$T1 - $T0" | bc) s"
grep WARNING gpurun_out/r06_bench_driver_cmd.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06_bench_driver_cmd.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["config"]["sort_mode"], d["config"]["frames_in_flight"], d["config"]["one_frame_at_a_time_ms"], d["config"]["headline_reason"])
print({m:x["ms_per_step"] for m,x in d["modes"].items()})
print(d["parity_vs_oracle"]["visible_in_flight"])
print(d["sort_mode_cross_check"]["in_flight"])
print(d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["valu"]["frac_of_spec"], d["roofline"]["whole_frame"], d["vs_baseline"], d["vs_baseline_by_mode"])
PY

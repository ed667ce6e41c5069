#!/bin/bash
# The validation call of a round (on the GPU box, from the repo root): the whole -m gpu suite, smoke(), then the previous round's build against
# HEAD (scripts/ab_rounds.sh).  Outputs under gpurun_out/.
export PYTHONPATH=$PWD
O=$PWD/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash scripts/ab_rounds.sh "${1:-$PWD/unitygaussiansplatting_amd/variants/r04.so}" C2 C3 C4 C2d 2>&1 | tail -20

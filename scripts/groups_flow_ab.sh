#!/bin/bash
# (GPU) two against three job groups, in the default bench flow (the H2D-inclusive pass first) and resident-only: resident ms per step, one box, interleaved
REPS=${1:-3}
for rep in $(seq 1 $REPS); do
  for g in 2 3; do
    echo -n "groups $g full flow: "
    FLORIA_HIP_GROUPS=$g python bench.py --steps 6 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --eps2 0 --h2d-steps 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
    echo -n "groups $g resident-only: "
    FLORIA_HIP_GROUPS=$g python bench.py --steps 6 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --eps2 0 --resident-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done

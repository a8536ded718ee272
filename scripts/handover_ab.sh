#!/bin/bash
# NOTE: the FLORIA_HIP_HANDOVER knob this script drives was measured, rejected and removed (profiles/r05_handover_ab.txt); the script is kept as the record of how it was measured.
# (GPU) job groups x handover (FLORIA_HIP_HANDOVER: the groups' beam launches of a stage hand the chip over instead of racing for it): resident ms per step, REPS runs each, interleaved
REPS=${1:-4}
for rep in $(seq 1 $REPS); do
  for cfg in "2 0" "3 0" "3 1" "3 2" "2 1" "2 2" "4 1"; do
    set -- $cfg
    echo -n "groups $1 handover $2: "
    FLORIA_HIP_GROUPS=$1 FLORIA_HIP_HANDOVER=$2 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-only --eps2 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step_resident'], d['roofline']['beam_union_ms_per_step'])"
  done
done

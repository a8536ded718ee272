#!/bin/bash
# NOTE: the FLORIA_HIP_GROUP_SKEW knob this script drives was measured and removed (profiles/r05_groups_ab.txt); the script is kept as the record of how it was measured.
# (GPU) three job groups of unequal size (FLORIA_HIP_GROUP_SKEW): does breaking the symmetry pin one interleaving?  resident ms per step, the FULL default bench flow (H2D pass first)
REPS=${1:-4}
for rep in $(seq 1 $REPS); do
  for sk in 0 2 4 -2 6; do
    echo -n "skew $sk: "
    FLORIA_HIP_GROUP_SKEW=$sk python bench.py --steps 6 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --eps2 0 --h2d-steps 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['beam_union_ms_per_step'])"
  done
done

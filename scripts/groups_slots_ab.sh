#!/bin/bash
# (GPU) job groups x resident beam waves per launch (bench.py --slots): does an equal share of the wave slots per group make three groups deterministic?  resident ms per step
REPS=${1:-3}
for rep in $(seq 1 $REPS); do
  for cfg in "2 0" "3 0" "3 1366" "3 2048" "3 2731" "2 2048" "2 3072" "4 1024" "4 2048"; do
    set -- $cfg
    echo -n "groups $1 slots $2: "
    FLORIA_HIP_GROUPS=$1 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-only --eps2 0 $( [ $2 != 0 ] && echo --slots $2 ) 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['beam_union_ms_per_step'])"
  done
done

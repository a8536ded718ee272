#!/bin/bash
# usage (GPU box): scripts/pipe_groups_ab.sh -> H2D-inclusive ms of config-4 batches with the chunk groups merged into fewer job groups
for rep in 1 2; do for N in 500 1000 2000; do for PG in 0 1 2 3; do
  echo -n "[$rep] contigs=$N pipe_groups=$PG: "
  FLORIA_HIP_PIPE_GROUPS=$PG python bench.py --contigs $N --steps 4 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['parallelism'][-30:])"
done; done; done

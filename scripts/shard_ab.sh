#!/bin/bash
# usage (GPU box): scripts/shard_ab.sh  -> resident ms of config-4 shards under the speculative-stage variants
for rep in 1 2; do for D in 1 2 4 8; do for N in 125 250 500; do
  echo -n "[$rep] gate_div=$D contigs=$N: "
  FLORIA_HIP_SPEC_GATE_DIV=$D FLORIA_HIP_SPECULATE=1 python bench.py --contigs $N --steps 5 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(d['value_resident'], d['ms_per_step_resident'], 'beam', k['beam'], 'opt', k['optimize'])"
done; done; done

#!/bin/bash
# usage (GPU box): scripts/spec_big.sh -> resident ms of config-4 batches with / without speculative ploidy stages and 1 / 2 job groups
for rep in 1 2; do for N in 375 500 750; do for S in 0 1; do for G in 1 2; do
  echo -n "[$rep] contigs=$N speculate=$S groups=$G: "
  FLORIA_HIP_GROUPS=$G FLORIA_HIP_SPECULATE=$S python bench.py --contigs $N --steps 4 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(d['value_resident'], d['ms_per_step_resident'], 'beam', k['beam'], 'opt', k['optimize'], 'groups', k['job_groups'], 'width', k['ploidies_per_stage'])"
done; done; done; done

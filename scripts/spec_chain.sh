#!/bin/bash
# usage (GPU box): scripts/spec_chain.sh "500 1000 2000" -> resident ms, one ploidy per stage against the chained speculative stage (speculate = 4)
for N in ${1:-500 750 1000 1500 2000}; do for S in 0 4; do for G in 1 2; do
  echo -n "contigs=$N speculate=$S groups=$G: "
  FLORIA_HIP_GROUPS=$G FLORIA_HIP_SPECULATE=$S timeout 120 python bench.py --contigs $N --steps 4 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(d['value_resident'], d['ms_per_step_resident'], 'beam', k['beam'], 'opt', k['optimize'], 'groups', k['job_groups'], 'width', k['ploidies_per_stage'])" 2>&1 | tail -1
done; done; done

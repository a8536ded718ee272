#!/bin/bash
# usage (GPU box): scripts/spec_modes.sh "250 375 500 625 750" -> resident ms under speculate = 0 / 1 / 2
for N in ${1:-250 375 500 625 750}; do for S in 0 1 2; do
  echo -n "contigs=$N speculate=$S: "
  FLORIA_HIP_SPECULATE=$S python bench.py --contigs $N --steps 4 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(d['value_resident'], d['ms_per_step_resident'], 'blocks', d['config']['blocks_this_rank'], 'groups', k['job_groups'], 'width', k['ploidies_per_stage'])"
done; done

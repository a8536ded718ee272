#!/bin/bash
# (GPU) usage: scripts/opt_order_ab.sh [contigs ...]  — optimise kernel: reads visited longest first (default) against block order (FLORIA_HIP_OPT_BLOCK_ORDER=1), resident bench, twice, interleaved
for rep in 1 2; do for N in ${@:-2000}; do for V in 1 0; do
  echo -n "[$rep] contigs=$N block_order=$V: "
  if [ $V = 1 ]; then export FLORIA_HIP_OPT_BLOCK_ORDER=1; else unset FLORIA_HIP_OPT_BLOCK_ORDER; fi
  timeout 300 python bench.py --contigs $N --steps 6 --warmup 2 --cpu-sample 0 --check 4 --pipeline 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('incl', d['ms_per_step'], 'resident', d['ms_per_step_resident'], 'beam', k['beam'], 'opt', k['optimize'], 'mismatches', d['spot_check']['mismatches'])"
done; done; done

#!/bin/bash
# (GPU) usage: scripts/tail_ab.sh [extra bench flags]  — A/B of Knobs::tail_overlap (the last ploidy's beam launch beside the optimise launch below it) on the default bench, twice, interleaved
for rep in 1 2; do for v in 0 1; do
  echo -n "[$rep] tail_overlap $v: "
  FLORIA_HIP_TAIL_OVERLAP=$v timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --check 4 --pipeline 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(d['value'], d['ms_per_step'], 'resident', d['value_resident'], d['ms_per_step_resident'], 'beam', k['beam'], 'opt', k['optimize'], 'loop', k['launch_loop_wall'], 'mismatches', d['spot_check']['mismatches'])"
done; done

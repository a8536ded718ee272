#!/bin/bash
# (GPU) full config 4, resident: one ploidy per stage against {1}{2}{3}{4..P} (speculate 3), and the child-sum shuffle variant; twice, interleaved
for rep in 1 2; do for sp in 0 3; do
  echo -n "[$rep] speculate $sp: "
  FLORIA_HIP_SPECULATE=$sp timeout 600 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --check 4 --pipeline 0 --resident-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(d['value_resident'], d['ms_per_step_resident'], 'beam', k['beam'], 'opt', k['optimize'], 'loop', k['launch_loop_wall'])"
done; done

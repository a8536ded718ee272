#!/bin/bash
# (GPU) the two timings a kernel change is judged by: the canonical resident step with TWO job groups (three are bimodal: profiles/r05_groups_ab.txt), twice, and the reference-arithmetic call, twice
for i in 1 2; do FLORIA_HIP_GROUPS=2 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('canonical, two groups:', d['value'], d['ms_per_step'], 'optimise ms', d['roofline']['kernel_ms_per_step']['optimize'])"; done
for i in 1 2; do python scripts/arith_timing.py 2000 0.04 2>&1 | tail -1; done

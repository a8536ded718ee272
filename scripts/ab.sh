#!/bin/bash
# usage (on the GPU box): scripts/ab.sh "<EXTRA flags A>" "<EXTRA flags B>" ...   -> resident blocks/s and kernel ms for each variant, twice, interleaved
for rep in 1 2; do
for v in "$@"; do
  make -C floria_amd/csrc -B EXTRA="$v" libfloria_hip.so > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo -n "[$rep] '$v': "
  python bench.py --steps 5 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-only --eps2 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(d['value_resident'], d['ms_per_step_resident'], 'beam', k['beam'], 'opt', k['optimize'], 'loop', k['launch_loop_wall'])"
done; done
make -C floria_amd/csrc -B libfloria_hip.so > /dev/null 2>&1

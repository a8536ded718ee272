#!/bin/bash
# usage: scripts/variants_bench.sh "<EXTRA flags>" ...   (builds each variant on the GPU box and prints the bench line's key numbers)
for v in "$@"; do
  make -C floria_amd/csrc -B EXTRA="$v" libfloria_hip.so > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "== $v"
  for G in ${GROUPS_LIST:-1 2}; do FLORIA_HIP_GROUPS=$G python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"; done
done
make -C floria_amd/csrc -B libfloria_hip.so > /dev/null 2>&1

#!/bin/bash
# (GPU) usage: scripts/arith_ab.sh "<flags A>" "<flags B>" ...  — rebuild the library with each flag set and time the reference-arithmetic mode (scripts/arith_timing.py 2000 0.04), twice each, interleaved
for rep in 1 2; do
for v in "$@"; do
  make -C floria_amd/csrc -B EXTRA="$v" libfloria_hip.so > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo -n "[$v] "; python scripts/arith_timing.py 2000 0.04 2>&1 | tail -1
done
done
make -C floria_amd/csrc -B libfloria_hip.so > /dev/null 2>&1

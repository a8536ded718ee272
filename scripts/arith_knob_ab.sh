#!/bin/bash
# (GPU) usage: scripts/arith_knob_ab.sh "<knob=value ..>" "<..>" ...  — time the reference-arithmetic mode (scripts/arith_timing.py 2000 0.04) under each set of context options ("-" = none), twice each, interleaved
for rep in 1 2; do
for v in "$@"; do
  [ "$v" = "-" ] && kv="" || kv="$v"
  echo -n "[$v] "; python scripts/arith_timing.py 2000 0.04 -1 0 $kv 2>&1 | tail -1
done
done

#!/bin/bash
# usage: variants.sh "<EXTRA flags variant 1>" "<variant 2>" ...   (run on the GPU box)
for v in "$@"; do
  make -C floria_amd/csrc -B EXTRA="$v" libfloria_hip.so > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "== $v"
  timeout 600 python scripts/quick_batch.py 512 1 4 0 2>&1 | grep -E "iter 2|parity"
done

#!/bin/bash
# (GPU) rebuilds the library with each flag set and runs the lean-kernel parity test: which mechanism breaks parity?
for v in "$@"; do
  make -C floria_amd/csrc -B EXTRA="$v" libfloria_hip.so > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "== '$v'"
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "lean_kernel" 2>&1 | grep -E "AssertionError: seed|passed|failed" | head -8
done
make -C floria_amd/csrc -B libfloria_hip.so > /dev/null 2>&1

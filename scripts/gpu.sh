#!/bin/bash
# usage: scripts/gpu.sh <timeout s> '<command>'  — gpurun with retries while no GPU slot is free (exit code 3 charges nothing)
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3

#!/bin/bash
# (GPU) usage: scripts/variants_so_ab.sh "<command>" name ...  — libraries prebuilt on the build host as floria_amd/csrc/variants/libfloria_hip_<name>.so (flags of one's choice; "base" = the
# tree's own library) take the place of libfloria_hip.so in turn, "<command>" runs twice under each, interleaved
D=floria_amd/csrc
cp $D/libfloria_hip.so $D/variants/libfloria_hip_base.so
cmd=$1; shift
for rep in 1 2; do
for v in "$@"; do
  cp $D/variants/libfloria_hip_$v.so $D/libfloria_hip.so
  echo -n "[$v] "; bash -c "$cmd" 2>&1 | tail -1
done
done
cp $D/variants/libfloria_hip_base.so $D/libfloria_hip.so

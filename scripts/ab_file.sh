#!/bin/bash
# (GPU) usage: scripts/ab_file.sh [file under floria_amd/csrc, default beam_slab_kernel.h]
# A/B of the working copy of that file against a previous version placed in scripts/tmp_prev/ (git show HEAD:... > scripts/tmp_prev/beam_slab_kernel.h; not committed):
# file-level variants for changes that no -D switch guards.  Resident bench, twice, interleaved.
F=${1:-beam_slab_kernel.h}
cp floria_amd/csrc/$F /tmp/new_$F
for rep in 1 2; do for v in prev new; do
  if [ $v = prev ]; then cp scripts/tmp_prev/$F floria_amd/csrc/$F; else cp /tmp/new_$F floria_amd/csrc/$F; fi
  make -C floria_amd/csrc -B libfloria_hip.so > /dev/null 2>&1 || { echo "BUILD FAILED $v"; continue; }
  echo -n "[$rep] $v: "
  python bench.py --steps 5 --warmup 2 --cpu-sample 0 --check 4 --pipeline 0 --resident-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(d['value_resident'], d['ms_per_step_resident'], 'beam', k['beam'], 'opt', k['optimize'], 'loop', k['launch_loop_wall'])"
done; done
cp /tmp/new_$F floria_amd/csrc/$F; make -C floria_amd/csrc -B libfloria_hip.so > /dev/null 2>&1

#!/bin/bash
# (GPU) beam steps per second against the number of resident jobs, one job group, for each beam path given (slab lean): the lone-wave step time and the loaded one
for v in "$@"; do
  for slots in 256 1024 2048 3072 0; do
    echo -n "$v slots $slots: "
    FLORIA_HIP_BEAM=$v FLORIA_HIP_GROUPS=1 timeout 900 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --check 0 --pipeline 0 --resident-only --slots $slots 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']; print('ms/step', d['ms_per_step_resident'], 'beam ms', k['kernel_ms_per_step']['beam'], 'beam steps/s %.2fM'%(k['beam_steps_per_s']/1e6))"
  done
done

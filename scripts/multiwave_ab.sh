#!/bin/bash
# (GPU) VERDICT r5 #1, measured before building: would more waves per (block, ploidy) job shorten an under-filled launch?
#  1. phase A of the beam step with 1 / 1/2 / 1/4 of its lanes per live slab (variants a1, a2: -DFLORIA_MW_A_SHIFT=1|2, built on the build host): a phase bound by its
#     lanes gets 2x / 4x longer, and would get shorter with the lanes of more waves; a phase that is one dependent round trip does not move;
#  2. the 250-contig shard (the per-GPU share at 8 GPUs) and the lone-wave regime (--slots 256) of the full job under each;
#  3. per-phase cycle counters (-DFLORIA_PROF) of the shard: what share of a step's chain the lane-parallel phases (staging, distance, read-modify-write) are.
D=floria_amd/csrc
cp $D/libfloria_hip.so /tmp/libfloria_hip_base.so
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']; print('ms/step', d['ms_per_step_resident'], 'beam ms', k['kernel_ms_per_step']['beam'], 'optimise ms', k['kernel_ms_per_step']['optimize'], 'beam steps/s %.2fM'%(k['beam_steps_per_s']/1e6))"; }
for rep in 1 2; do
for v in base a1 a2; do
  [ $v = base ] && cp /tmp/libfloria_hip_base.so $D/libfloria_hip.so || cp $D/variants/libfloria_hip_$v.so $D/libfloria_hip.so
  echo -n "[$rep] $v  shard 250 contigs: "; python bench.py --contigs 250 --steps 10 --warmup 3 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2>/dev/null | line
  echo -n "[$rep] $v  full job, one lone wave per CU (--slots 256, one group): "; FLORIA_HIP_GROUPS=1 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --check 0 --pipeline 0 --resident-only --slots 256 2>/dev/null | line
done
done
cp $D/variants/libfloria_hip_prof.so $D/libfloria_hip.so
echo "--- FLORIA_PROF, shard of 250 contigs (every launch of one resident S1 call, all ploidies):"
python bench.py --contigs 250 --steps 1 --warmup 0 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2>&1 | python scripts/multiwave_prof_fmt.py
echo "--- FLORIA_PROF, full job with one lone wave per CU (--slots 256):"
FLORIA_HIP_GROUPS=1 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --pipeline 0 --resident-only --slots 256 2>&1 | python scripts/multiwave_prof_fmt.py
cp /tmp/libfloria_hip_base.so $D/libfloria_hip.so

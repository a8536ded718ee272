#!/bin/bash
# (GPU) cycle counters of the beam and optimise kernels' phases, canonical against the reference's arithmetic, one job group (so that phases are not interleaved across groups).
# needs floria_amd/csrc/variants/libfloria_hip_prof.so (-DFLORIA_PROF, built on the build host).   usage: scripts/arith_phase_prof.sh [contigs = 2000]
D=floria_amd/csrc
cp $D/libfloria_hip.so /tmp/libfloria_hip_base.so; cp $D/variants/libfloria_hip_prof.so $D/libfloria_hip.so
python scripts/arith_timing.py ${1:-2000} 0.04 -1 0 groups=1 2>&1 | grep -E "^\[prof\]|^arith" | python -c "
import sys,re
rows=[]
for l in sys.stdin:
    if l.startswith('arith'): print(l.strip()); continue
    d={int(a):float(b) for a,b in re.findall(r'(\d+):([0-9.]+)M',l)}
    rows.append(d)
# every S1 call prints one cumulative-per-call line; calls: arith0 warm, arith0 timed, arith1 warm, arith1 timed
for name,d in zip(('arith 0 (warm-up)','arith 0','arith 1 (warm-up)','arith 1'),rows):
    g=lambda i: d.get(i,0)/1e3
    print('%-18s beam Gcyc: stage %.1f A %.1f B %.1f M1 %.1f M2 %.1f adds %.1f tail %.1f | opt Gcyc (thread 0 per workgroup): build %.2f stats0 %.2f dist %.2f cand %.2f sort %.2f serial %.2f moves %.2f stats %.2f undo %.2f final %.2f | arith: keys-clear %.2f atomicMin %.2f keys %.2f sort %.2f replay+walk %.2f walk(p0) %.2f'%(name,g(16),g(17),g(18),g(19),g(20),g(21),g(22),g(0),g(1),g(2),g(3),g(4),g(5),g(6),g(7),g(8),g(62),g(14),g(15),g(10),g(11),g(12),g(13)))
"
cp /tmp/libfloria_hip_base.so $D/libfloria_hip.so

#!/bin/bash
# (GPU) cycle counters of the beam kernel's phases, canonical against the reference's arithmetic: needs floria_amd/csrc/variants/libfloria_hip_prof.so (built with -DFLORIA_PROF on the build host)
D=floria_amd/csrc
cp $D/libfloria_hip.so $D/variants/libfloria_hip_base.so; cp $D/variants/libfloria_hip_prof.so $D/libfloria_hip.so
python scripts/arith_timing.py ${1:-500} 0.04 2>&1 | grep -E "^\[prof\]|^arith" | python -c "
import sys,re
for l in sys.stdin:
    if l.startswith('arith'): print(l.strip()); continue
    d={int(a):float(b) for a,b in re.findall(r'(\d+):([0-9.]+)M',l)}
    g=lambda i: d.get(i,0)
    print('  beam phases (Gcyc): stage %.1f A %.1f B %.1f M1 %.1f M2 %.1f adds %.1f tail %.1f | steps %.2fM nlive %.1fM nstates %.1fM cells %.1fM code items %.1fM'%(g(16)/1e3,g(17)/1e3,g(18)/1e3,g(19)/1e3,g(20)/1e3,g(21)/1e3,g(22)/1e3,g(9) and 0 or 0,g(13),g(15),g(9),g(54)))"
cp $D/variants/libfloria_hip_base.so $D/libfloria_hip.so

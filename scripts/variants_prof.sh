#!/bin/bash
for v in "$@"; do
  make -C floria_amd/csrc -B EXTRA="$v -DFLORIA_PROF" libfloria_hip.so > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "== $v"
  python bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --pipeline 0 --resident-only ${SLOTS:+--slots $SLOTS} 2>&1 | grep -E "^\[prof\]" | tail -1 | python -c "
import sys,re
import collections
d=collections.defaultdict(float)
for l in sys.stdin:
    for a,b in re.findall(r'(\d+):([0-9.]+)M',l): d[int(a)]+=float(b)
print('beam phases (Gcyc): stage %.0f A %.0f B %.0f M1 %.0f M2 %.0f adds %.0f tail %.0f | B1 %.0f | opt build %.0f dist %.0f'%(d[16]/1e3,d[17]/1e3,d[18]/1e3,d[19]/1e3,d[20]/1e3,d[21]/1e3,d[22]/1e3,d[23]/1e3,d[0]/1e3,d[2]/1e3)); print('opt phases (Gcyc, thread 0 of each workgroup): build %.1f stats0 %.1f dist %.1f cand %.1f sort %.1f serial %.1f moves %.1f stats %.1f undo %.1f final %.1f'%tuple(d[i]/1e3 for i in list(range(9))+[62])); print('beam wave-seconds per ploidy launch 2..5: %.1f %.1f %.1f %.1f'%tuple(d[28+i]*1e6/1e8 for i in range(4))); print('sum over steps: nlive %.1fM nin %.1fM nstates %.1fM cells %.1fM'%(d[13],d[14],d[15],d[9])); print('children that pass %.1fM  pushed %.1fM  popped %.1fM'%(d[10],d[11],d[12])); print('copies %.1fM covering %.1fM positions; distinct new versions %.1fM, add items (version x cell) %.1fM; zeroed (slab x position) %.1fM; steps with window exits %.1fM'%(d[33],d[32],d[36],d[34],d[35],d[37])); print('next-live slabs %.1fM with id>=8 %.2fM >=16 %.2fM >=32 %.2fM; window sum %.1fM steps with window>128 %.2fM >256 %.2fM >512 %.2fM; steps with add items>64 %.2fM >128 %.2fM >256 %.2fM; level-2 prune steps %.2fM (exact log-sum-exp in %.2fM); general-path steps %.2fM; bulk steps with as many survivors as states, no copy and an unchanged live count %.2fM, of which with the heap kept as it is %.2fM'%(d[38],d[39],d[40],d[41],d[45],d[42],d[43],d[44],d[48],d[49],d[50],d[51],d[53],d[52],d[60],d[61])); print('lean kernel: steps %.1fM bulk %.1fM cached (rows) %.1fM; add items from the LDS prefetch %.1fM, direct %.1fM; bookkeeping runs %.1fM'%(d[54],d[55],d[56],d[57],d[58],d[59])); print('beam waves %.0f  sum wave wall ticks(100MHz) %.1fM  sum core cyc %.1fM -> core clock %.2f GHz'%(d[26]*1e6,d[24],d[25],d[25]/max(d[24],1e-9)*0.1))"
  make -C floria_amd/csrc -B EXTRA="$v" libfloria_hip.so > /dev/null 2>&1
  python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done

#!/bin/bash
# (GPU) per-dispatch kernel durations of one resident S1 call in the reference-arithmetic mode: which ploidy's launches the time goes to
# usage: scripts/arith_trace.sh [contigs = 2000] [eps = 0.04]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/arith_trace
( cd $R && rocprofv3 --kernel-trace --output-format csv -d /tmp/arith_trace -- python scripts/arith_timing.py ${1:-2000} ${2:-0.04} ) 2>&1 | grep -E "^arith"
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/arith_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = None
# the last S1 call = the last run of beam/optimize kernels; print the last 40 dispatches of those families
sel = [r for r in rows if 'beam' in r['Kernel_Name'] or 'optimize' in r['Kernel_Name']]
for r in sel[-int(__import__("os").environ.get("TRACE_ROWS", "24")):]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t0 is None: t0 = s
    name = r['Kernel_Name']
    name = name[name.find('fl::'):][:70]
    print(f"{(s - t0) / 1e6:9.2f} ms  +{(e - s) / 1e6:8.2f} ms  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>8}  {name}")
PY

#!/bin/bash
# usage (GPU box): scripts/insts.sh "<EXTRA flags A>" "<EXTRA flags B>" ...
# per variant: VALU / SALU instructions of the beam and optimise kernels per S1 call and per beam step (rocprofv3 --pmc, its own pass),
# then the resident bench (kernel ms per step) without counters
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  make -C $R/floria_amd/csrc -B EXTRA="$v" libfloria_hip.so > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  echo "== '$v'"
  O=$R/gpurun_out/insts_tmp; rm -rf $O; mkdir -p $O
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O -o x -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --pipeline 0 --resident-only > $O/log 2>&1)
  python - <<PY
import csv, glob, collections, json
fs=glob.glob("$O/**/*counter_collection.csv", recursive=True)
tot=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    k=r["Kernel_Name"]
    fam="beam" if "beam_" in k else "optimize" if "optimize" in k else None
    if fam: tot[fam][r["Counter_Name"]]+=float(r["Counter_Value"])
line=[l for l in open("$O/log") if l.startswith("{")]
steps=None
if line:
    d=json.loads(line[-1]); k=d["roofline"]; steps=k["beam_steps_per_s"]*k["kernel_ms_per_step"]["beam"]*1e-3
calls=2.0   # one warm-up and one timed S1 call in the counter pass
for fam,v in tot.items():
    s=" ".join("%s %.2fG"%(n[9:],x/calls/1e9) for n,x in sorted(v.items()))
    per=""
    if fam=="beam" and steps: per=" | per beam step: VALU %.0f SALU %.0f LDS %.0f VMEM %.1f+%.1f (%.2fM steps)"%(v["SQ_INSTS_VALU"]/calls/steps, v["SQ_INSTS_SALU"]/calls/steps, v["SQ_INSTS_LDS"]/calls/steps, v["SQ_INSTS_VMEM_RD"]/calls/steps, v["SQ_INSTS_VMEM_WR"]/calls/steps, steps/1e6)
    print(fam, "per S1 call:", s, per)
PY
  for rep in 1 2; do
  python $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('  resident', d['value_resident'], d['ms_per_step_resident'], 'beam', k['beam'], 'opt', k['optimize'], 'loop', k['launch_loop_wall'])"
  done
  FLORIA_HIP_GROUPS=1 python $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print('  one group', d['value_resident'], d['ms_per_step_resident'], 'beam', k['beam'], 'opt', k['optimize'], 'loop', k['launch_loop_wall'])"
done
make -C $R/floria_amd/csrc -B libfloria_hip.so > /dev/null 2>&1

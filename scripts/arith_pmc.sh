#!/bin/bash
# (GPU) SQ counters of the kernels of the reference-arithmetic mode (one rocprofv3 --pmc pass per counter set; scripts/arith_timing.py <contigs> 0.04)
# usage: scripts/arith_pmc.sh [contigs = 1000] "<COUNTERS set 1>" "<COUNTERS set 2>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-1000}; shift
for C in "$@"; do
  O=/tmp/arith_pmc; rm -rf $O; mkdir -p $O
  (cd $R && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O -o x -- python scripts/arith_timing.py $N 0.04 > $O/log 2>&1)
  python - <<PY
import csv, glob, collections
fs=glob.glob("$O/**/*counter_collection.csv", recursive=True)
tot=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    k=r["Kernel_Name"].split("(")[0].replace("void fl::","")
    if "true>" in k and ("beam" in k or "optimize" in k): tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in sorted(tot.items()): print(k, " ".join("%s=%.4gG"%(n,x/1e9) for n,x in sorted(v.items())))
PY
done

#!/bin/bash
# usage (GPU box): scripts/pmc.sh "COUNTER1 COUNTER2 ..." [extra bench args]   -> per-kernel-family totals of the counters (one rocprofv3 --pmc pass, resident bench)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_tmp; rm -rf $O; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $O -o x -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --pipeline 0 --resident-only ${@:2} > $O/log 2>&1)
python - <<PY
import csv, glob, collections
fs=glob.glob("$O/**/*counter_collection.csv", recursive=True)
tot=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    k=r["Kernel_Name"].split("(")[0].replace("void fl::","")
    if "beam" in k or "optimize" in k: tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in sorted(tot.items()): print(k, " ".join("%s=%.4gG"%(n,x/1e9) for n,x in sorted(v.items())))
PY

#!/bin/bash
# (GPU) HBM traffic of the reference-arithmetic instances: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes over two resident S1 calls of
# BASELINE config 4 at -e 0.04 in arith = 1 (scripts/arith_timing.py), per kernel instance and per call, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md.
# usage: scripts/arith_pmc_round.sh <tag>   -> gpurun_out/<tag>_pmc_arith.json
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for C in FETCH_SIZE WRITE_SIZE; do
  O=/tmp/arith_pmc_$C; rm -rf $O; mkdir -p $O
  (cd $R && export TMPDIR=/tmp && ARITH_MODES=1 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O -o x -- python scripts/arith_timing.py 2000 0.04 > $O/log 2>&1)
done
cd $R
python - <<PY
import csv, glob, collections, json, hashlib
def agg(d, name):
    tot=collections.defaultdict(float); n=collections.defaultdict(set)
    for r in csv.DictReader(open(glob.glob(d+"/**/*counter_collection.csv", recursive=True)[0])):
        if r["Counter_Name"]==name:
            k=r["Kernel_Name"].split("(")[0].replace("void fl::",""); tot[k]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k:(v,len(n[k])) for k,v in tot.items()}
f=agg("/tmp/arith_pmc_FETCH_SIZE","FETCH_SIZE"); w=agg("/tmp/arith_pmc_WRITE_SIZE","WRITE_SIZE")
calls=2.0
out={"workload":"config4, 2000 contigs, -e 0.04, arith = 1 (the reference's running sums), resident; two S1 calls per pass","note":"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KiB); the read side doubled per the gfx950 note (upper bound for scattered accesses)","kernels":{}}
for k in f:
    if "beam" in k or "optimize" in k or "cell_order" in k:
        fk,nl=f[k]; wk,_=w.get(k,(0,nl))
        out["kernels"][k]={"launches":nl,"fetch_kib":fk,"write_kib":wk,"hbm_bytes_per_launch_corrected":(2*fk+wk)*1024/nl,"hbm_bytes_per_call_corrected":(2*fk+wk)*1024/calls}
out["hbm_bytes_per_call"]={fam: sum(v["hbm_bytes_per_call_corrected"] for k,v in out["kernels"].items() if key in k) for fam,key in (("beam","beam_"),("optimize","optimize"))}
h=hashlib.sha256()
for fn in sorted(glob.glob("floria_amd/csrc/*.h")+glob.glob("floria_amd/csrc/*.hip")): h.update(open(fn,"rb").read())
out["kernel_sources_sha16"]=h.hexdigest()[:16]
json.dump(out, open("gpurun_out/${TAG}_pmc_arith.json","w"), indent=1)
print(json.dumps(out["hbm_bytes_per_call"]), len(out["kernels"]), "kernel instances")
PY

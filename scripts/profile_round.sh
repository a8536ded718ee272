#!/bin/bash
# Run on the GPU box: kernel-trace stats + the two HBM-traffic PMC passes of the default bench workload.
# usage: scripts/profile_round.sh r01b
set -u
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o $TAG -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o $TAG -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/bench_write.log 2>&1
cd $R
python - <<PY
import csv, json, collections, glob
O="$O"; TAG="$TAG"
def agg(path, name):
    rows=list(csv.DictReader(open(glob.glob(path+"/*counter_collection.csv")[0])))
    tot=collections.defaultdict(float); n=collections.defaultdict(set)
    for r in rows:
        if r["Counter_Name"]==name:
            k=r["Kernel_Name"].split("(")[0]; tot[k]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k:(v,len(n[k])) for k,v in tot.items()}
f=agg(O+"/pmc_fetch","FETCH_SIZE"); w=agg(O+"/pmc_write","WRITE_SIZE")
out={"workload":"config4","contigs_per_rank":2000,"note":"rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (bench.py --steps 1 --warmup 0); FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE under-reports wide coalesced streams by 2x (MI355X_MICROARCH.md §HBM), so the read side is doubled (upper bound for this scattered 16-B access pattern)","kernels":{}}
for k in f:
    if "beam" in k or "optimize" in k:
        fk,nl=f[k]; wk,_=w.get(k,(0,nl))
        out["kernels"][k]={"launches":nl,"fetch_kib":fk,"write_kib":wk,"hbm_bytes_per_launch_raw":(fk+wk)*1024/nl,"hbm_bytes_per_launch_corrected":(2*fk+wk)*1024/nl}
bk=[k for k in out["kernels"] if "beam_slab" in k]
if bk: out["hbm_bytes_per_launch"]=out["kernels"][bk[0]]["hbm_bytes_per_launch_corrected"]
json.dump(out, open(O+"/pmc_summary.json","w"), indent=1)
print(json.dumps(out)[:600])
PY
cp $O/stats/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
tail -1 $O/bench_stats.log > $O/bench.json

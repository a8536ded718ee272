#!/bin/bash
# Run on the GPU box: kernel-trace stats + the two HBM-traffic PMC passes of the default bench workload.
# usage: scripts/profile_round.sh r01b
set -u
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --check 0 --pipeline 0 --resident-only > $O/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o $TAG -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --pipeline 0 --resident-only > $O/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o $TAG -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --pipeline 0 --resident-only > $O/bench_write.log 2>&1
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
out={"workload":"config4","contigs_per_rank":2000,"note":"rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (bench.py --steps 1 --warmup 0 --resident-only: one warm-up and one timed S1 call over the resident batch); FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE under-reports wide coalesced streams by 2x (MI355X_MICROARCH.md §HBM), so the read side is doubled (upper bound for this scattered 16-B access pattern)","kernels":{}}
for k in f:
    if "beam" in k or "optimize" in k:
        fk,nl=f[k]; wk,_=w.get(k,(0,nl))
        out["kernels"][k]={"launches":nl,"fetch_kib":fk,"write_kib":wk,"hbm_bytes_per_launch_raw":(fk+wk)*1024/nl,"hbm_bytes_per_launch_corrected":(2*fk+wk)*1024/nl}
bk=[k for k in out["kernels"] if "beam_slab" in k]
if bk:      # calls-weighted mean over the kernel's ploidy-specialised instances (bench.py's roofline.traffic)
    out["hbm_bytes_per_launch"]=sum(out["kernels"][k]["hbm_bytes_per_launch_corrected"]*out["kernels"][k]["launches"] for k in bk)/sum(out["kernels"][k]["launches"] for k in bk)
# per S1 call (= step): every launch of the two kernel families; the counter passes ran two S1 calls (one warm-up, one timed)
calls=2.0
out["hbm_bytes_per_step"]={fam: sum((2*v["fetch_kib"]+v["write_kib"])*1024 for k,v in out["kernels"].items() if key in k)/calls for fam,key in (("beam","beam_"),("optimize","optimize"))}
import hashlib
h=hashlib.sha256()
for fn in sorted(glob.glob("floria_amd/csrc/*.h")+glob.glob("floria_amd/csrc/*.hip")): h.update(open(fn,"rb").read())
out["kernel_sources_sha16"]=h.hexdigest()[:16]
json.dump(out, open(O+"/pmc_summary.json","w"), indent=1)
print(json.dumps(out)[:600])
PY
cp $O/stats/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
python - <<PY
import csv, json
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
fam={}
for key in ("beam_slab_kernel","optimize_kernel"):
    rs=[r for r in rows if key in r["Name"]]
    calls=sum(int(r["Calls"]) for r in rs); tot=sum(float(r["TotalDurationNs"]) for r in rs)
    fam[key]={"instances":[{"name":r["Name"].split("(")[0],"calls":int(r["Calls"]),"avg_ms":float(r["AverageNs"])/1e6} for r in rs],
              "calls":calls,"avg_launch_ms":tot/calls/1e6 if calls else None,"total_ms":tot/1e6}
json.dump({"note":"per-family aggregate of rocprofv3 --kernel-trace --stats (bench.py --steps 3 --warmup 1 --resident-only: the resident pass the roofline block is measured in): the ploidy-specialised instances of one kernel are separate rows in kernel_stats.csv; bench.py's roofline.avg_launch_ms is the calls-weighted mean over the beam_slab_kernel family","families":fam}, open("$O/kernel_family_stats.json","w"), indent=1)
print(json.dumps({k:(v["calls"],round(v["avg_launch_ms"],3)) for k,v in fam.items()}))
PY
grep '^{' $O/bench_stats.log | tail -1 > $O/bench.json
# ---- SQ counters (two more passes): VALU / SALU / LDS / VMEM instruction counts, busy and wait cycles per kernel --------------------
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_sq1 -o $TAG -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --pipeline 0 --resident-only > $O/bench_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU --output-format csv -d $O/pmc_sq2 -o $TAG -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --pipeline 0 --resident-only > $O/bench_sq2.log 2>&1
cd $R
python - <<PY
import csv, json, collections, glob
O="$O"
raw=collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(float); nl=collections.defaultdict(set)
for d in ("pmc_sq1","pmc_sq2"):
    fs=glob.glob(O+"/"+d+"/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0]
        if "beam" in k or "optimize" in k:
            raw[k][r["Counter_Name"]]+=float(r["Counter_Value"])
# kernel durations of the un-instrumented stats run (sum over launches of one bench step = total / 3 timed+warm steps is not needed: use per-launch avg * launches of one step)
st=glob.glob(O+"/stats/**/*kernel_stats.csv", recursive=True)
avg={}; calls={}
if st:
    for r in csv.DictReader(open(st[0])):
        k=r["Name"].split("(")[0]; avg[k]=float(r["AverageNs"])/1e6; calls[k]=int(r["Calls"])
out={"note":"rocprofv3 --pmc SQ_* (two passes, bench.py --steps 1 --warmup 0, config 4 full; kernels are serialised by the counter collection); SQ_*_CYCLES are quad-cycles per SIMD-wave accounting as in MI355X_MICROARCH.md; valu_busy_frac = SQ_ACTIVE_INST_VALU*4 / SQ_BUSY_CYCLES-normalised SIMD cycles (SQ_BUSY_CYCLES counts per SE: reported raw)","kernels":{},"raw":{k:dict(v) for k,v in raw.items()}}
for k,v in raw.items():
    if "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"]>0:
        out["kernels"][k]={"avg_launch_ms_stats_run":avg.get(k),"launches_stats_run":calls.get(k),
            "insts_valu":v.get("SQ_INSTS_VALU"),"insts_salu":v.get("SQ_INSTS_SALU"),"insts_lds":v.get("SQ_INSTS_LDS"),
            "insts_vmem_rd":v.get("SQ_INSTS_VMEM_RD"),"insts_vmem_wr":v.get("SQ_INSTS_VMEM_WR"),
            "wave_cycles_waiting_frac": round(v.get("SQ_WAIT_ANY",0)/v["SQ_WAVE_CYCLES"],3),
            "valu_active_over_wave_cycles": round(v.get("SQ_ACTIVE_INST_VALU",0)/v["SQ_WAVE_CYCLES"],4)}
kt=glob.glob(O+"/pmc_sq2/**/*kernel_trace.csv", recursive=True)
if kt:
    dur=collections.defaultdict(float)
    for r in csv.DictReader(open(kt[0])):
        dur[r["Kernel_Name"].split("(")[0]]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-9
    for k in out["kernels"]:
        a=raw[k].get("SQ_ACTIVE_INST_VALU")
        if a and dur.get(k):
            out["kernels"][k]["kernel_seconds_in_counter_pass"]=round(dur[k],5)
            out["kernels"][k]["valu_busy_frac"]=round(a*4/(dur[k]*2.37e9*1024),3)      # 4 cycles per wave instruction, 1024 SIMDs, 2.37 GHz measured core clock
        tc=raw[k].get("SQ_THREAD_CYCLES_VALU")
        if a and tc:
            out["kernels"][k]["lane_utilisation"]=round(tc/(64.0*a),3)                 # active lanes per VALU instruction cycle / 64
json.dump(out, open(O+"/sq_counters.json","w"), indent=1)
print(json.dumps(out["kernels"])[:900])
PY

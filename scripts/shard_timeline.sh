#!/bin/bash
# kernel / copy timeline of one resident S1 call on a shard of BASELINE config 4 (usage on the GPU box: scripts/shard_timeline.sh 250)
R=$GRAFT_REPO_ROOT
N=${1:-250}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/tl_shard$N; mkdir -p $O
FLORIA_HIP_HW_QUEUES=6 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python $R/bench.py --contigs $N --steps 1 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-only --resident-steps 1 > $O/bench.log 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("fl::", "").replace("void ", "")[:60], "q" + r.get("Queue_Id", "?")))
for f in glob.glob("$O/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""), ""))
rows.sort()
# the last S1 call = from the last block_reads_kernel<false> on
i0 = max(i for i, r in enumerate(rows) if r[2].startswith("block_reads_kernel<false>"))
t0 = rows[i0][0]
with open("$O/timeline.txt", "w") as out:
    for s, e, n, q in rows[i0:]:
        out.write(f"{(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f} ms  {(e - s) / 1e6:8.3f}  {q:4s} {n}\n")
PY
rm -rf $O/*/ $O/*.csv 2>/dev/null
tail -1 $O/bench.log | cut -c1-400

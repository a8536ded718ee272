"""Dev helper: print the kernel / copy timeline of the LAST S1 step in a rocprofv3 --kernel-trace (+ --memory-copy-trace) csv dir.
usage: scripts/timeline.py <dir with *_kernel_trace.csv [*_memory_copy_trace.csv]> [min_us=200]"""
import csv, glob, sys
d = sys.argv[1]
min_ns = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 200e3
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("fl::", "")[:60], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") , ""))
rows.sort()
# steps are separated by host-side gaps; take events after the last big gap following a flatten/block_reads start
starts = [i for i, r in enumerate(rows) if r[2].startswith("block_reads_kernel<false>") or r[2].startswith("void block_reads_kernel<false>")]
i0 = starts[-1] if starts else 0
# include copies shortly before
t0 = rows[i0][0]
k = i0
while k > 0 and t0 - rows[k - 1][0] < 5e6: k -= 1
t0 = rows[k][0]
for s, e, n, q in rows[k:]:
    if (e - s) > min_ns or n.startswith("COPY"):
        print(f"{(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f} ms  {(e - s) / 1e6:8.3f}  {q:4s} {n}")

#!/bin/bash
# kernel timeline of one bench step per group count: gpurun_out/trace_G<g>/..._kernel_trace.csv
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for G in "$@"; do
  O=$R/gpurun_out/trace_G$G
  mkdir -p $O
  FLORIA_HIP_GROUPS=$G rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 > $O/bench.log 2>&1
  python - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("beam", "optimize", "select"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last step: the second half of the launches
n = len(rows) // 2
rows = rows[n:]
t0 = int(rows[0]["Start_Timestamp"])
print("G=$G  kernels in the timed step:", len(rows))
for r in rows:
    nm = r["Kernel_Name"].split("(")[0].replace("void fl::", "")[:28]
    print("%-28s q=%s grid=%s  start %8.2f  end %8.2f  dur %7.2f ms" % (nm, r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
done

#!/bin/bash
# (GPU) kernel timeline (with hardware queue ids) of the last resident S1 step with tail_overlap = $1 (default 1): gpurun_out/tail_trace_<v>.txt
V=${1:-1}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/tail_trace_$V
mkdir -p $O
FLORIA_HIP_HW_QUEUES=6 FLORIA_HIP_TAIL_OVERLAP=$V timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --check 0 --pipeline 0 --resident-only > $O/bench.log 2>&1
python - <<PY > $R/gpurun_out/tail_trace_$V.txt
import csv, glob
f = glob.glob("$O/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("beam", "optimize", "select"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 2
rows = rows[n:]
t0 = int(rows[0]["Start_Timestamp"])
print("tail_overlap=$V  kernels in the last step:", len(rows))
for r in rows:
    nm = r["Kernel_Name"].split("(")[0].replace("void fl::", "")[:44]
    print("%-44s q=%s grid=%s  start %8.2f  end %8.2f  dur %7.2f ms" % (nm, r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
rm -rf $O/*/*.csv $O/*.csv 2>/dev/null
cat $R/gpurun_out/tail_trace_$V.txt

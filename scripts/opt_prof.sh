#!/bin/bash
# (GPU) usage: scripts/opt_prof.sh [extra flags]  — cycle shares of the optimise kernel's phases (-DFLORIA_PROF: thread 0 of every workgroup, summed over the resident S1 call of the bench)
make -C floria_amd/csrc -B EXTRA="-DFLORIA_PROF $1" libfloria_hip.so > /dev/null 2>&1 || { echo "BUILD FAILED"; exit 1; }
cat > /tmp/opt_prof_fmt.py <<'PY'
import sys, re, collections
d = collections.defaultdict(float)
for l in sys.stdin:
    for a, b in re.findall(r'(\d+):([0-9.]+)M', l): d[int(a)] += float(b)
print('opt phases (Gcyc, thread 0 of each workgroup): build %.2f stats0 %.2f dist %.2f cand %.2f sort %.2f serial %.2f moves %.2f stats %.2f undo %.2f final %.2f | sum %.2f' % (tuple(d[i] / 1e3 for i in list(range(9)) + [62]) + (sum(d[i] for i in list(range(9)) + [62]) / 1e3,)))
PY
for V in 0 1; do
  if [ $V = 1 ]; then export FLORIA_HIP_OPT_BLOCK_ORDER=1; else unset FLORIA_HIP_OPT_BLOCK_ORDER; fi
  echo -n "block_order=$V  "
  timeout 300 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2>&1 | grep -E "^\[prof\]" | tail -1 | python /tmp/opt_prof_fmt.py
done

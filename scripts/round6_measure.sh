#!/bin/bash
# (GPU) the committed evidence of round 6 in one call: rocprofv3 passes of the canonical step (profile_round.sh), FETCH / WRITE passes of the reference-arithmetic
# instances, requested bytes by stream (prebuilt -DFLORIA_PROF variant), shard proxies in both arithmetics, the default bench line.   usage: scripts/round6_measure.sh r06
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/measure_$TAG; mkdir -p $O
cd $R
scripts/profile_round.sh $TAG > $O/profile_round.log 2>&1
scripts/arith_pmc_round.sh $TAG > $O/arith_pmc.log 2>&1
# requested bytes by stream: the prebuilt prof variant takes the library's place for one call
D=floria_amd/csrc
if [ -f $D/variants/libfloria_hip_prof.so ]; then
  cp $D/libfloria_hip.so /tmp/libfloria_hip_base.so; cp $D/variants/libfloria_hip_prof.so $D/libfloria_hip.so
  python bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2> gpurun_out/traffic_streams_$TAG.err > gpurun_out/traffic_streams_$TAG.out
  cp /tmp/libfloria_hip_base.so $D/libfloria_hip.so
  sed -n '/^python - <<PY/,/^PY/p' scripts/traffic_streams.sh | sed '1d;$d' | sed "s/\$TAG/$TAG/g" > /tmp/ts.py && python /tmp/ts.py > $O/traffic_streams.log 2>&1
fi
for n in 250 500 1000; do python bench.py --contigs $n --steps 10 --warmup 3 --cpu-sample 0 --check 0 --pipeline 0 2>/dev/null | grep "^{" > $O/shard_$n.json; done
python bench.py --steps 10 --warmup 3 2>/dev/null | grep "^{" > $O/bench_default.json
scripts/occupancy_sweep.sh slab > $O/occupancy_sweep.txt 2>&1
ls -la $O
python -c "
import json
d=json.load(open('$O/bench_default.json'))
print(d['value_is'], d['value'], d['ms_per_step'], d['value_h2d_inclusive'], d['ms_per_step_h2d_inclusive'], d['roofline']['frac'], d['roofline']['traffic'], d['second_pass'], d['gpu_over_cpu'])
for n in (250,500,1000):
    s=json.load(open('$O/shard_%d.json'%n)); print(n, s['ms_per_step_resident'], s['ms_per_step_h2d_inclusive'], s.get('second_pass',{}).get('reference_arithmetic'))
"

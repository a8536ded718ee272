#!/bin/bash
# (GPU) the committed evidence of a round: counter passes, occupancy sweep, shard lines, the default bench line.   usage: scripts/round_measure.sh r04
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/measure_$TAG; mkdir -p $O
cd $R
scripts/profile_round.sh $TAG > $O/profile_round.log 2>&1
scripts/occupancy_sweep.sh slab > $O/occupancy_sweep.txt 2>&1
for n in 250 500 1000; do python bench.py --contigs $n --steps 10 --warmup 3 --cpu-sample 0 --check 0 --pipeline 0 2>/dev/null | grep "^{" > $O/shard_$n.json; done
python bench.py --steps 10 --warmup 3 2>/dev/null | grep "^{" > $O/bench_default.json
ls -la $O

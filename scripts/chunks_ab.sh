#!/bin/bash
# usage (GPU box): scripts/chunks_ab.sh 4 5 6 8 ...   -> H2D-inclusive ms per step for each chunk count of the pipelined packed call, three rounds
for rep in 1 2 3; do
for c in "$@"; do
  echo -n "[$rep] upload_chunks=$c: "
  python bench.py --upload-chunks $c --steps 5 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done

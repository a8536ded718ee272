for rep in 1 2; do for N in 750 1000 1500 2000; do for G in 1 2 3; do
  echo -n "[$rep] contigs=$N groups=$G: "
  FLORIA_HIP_GROUPS=$G FLORIA_HIP_SPECULATE=0 python bench.py --contigs $N --steps 4 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(d['value_resident'], d['ms_per_step_resident'], 'beam', k['beam'], 'opt', k['optimize'])"
done; done; done

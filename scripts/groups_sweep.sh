for G in 2 3 4; do
  echo "== G=$G"
  FLORIA_HIP_GROUPS=$G python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done

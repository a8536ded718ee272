for rep in 1 2; do for T in 0 128 512 1024; do
  echo -n "[$rep] opt_threads=$T: "
  if [ $T = 0 ]; then unset FLORIA_HIP_OPT_THREADS; else export FLORIA_HIP_OPT_THREADS=$T; fi
  python bench.py --steps 5 --warmup 2 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(d['value_resident'], d['ms_per_step_resident'], 'beam', k['beam'], 'opt', k['optimize'], 'loop', k['launch_loop_wall'])"
done; done

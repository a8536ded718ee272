#!/bin/bash
# (GPU) the seeded sweeps beyond the test suite, one call: usage scripts/fuzz_round.sh <tag> [scale = 1]   -> gpurun_out/fuzz_<tag>/*.txt (copy what is to be judged into profiles/)
TAG=${1:-r06}; K=${2:-1}
O=gpurun_out/fuzz_$TAG; mkdir -p $O
python scripts/arith_fuzz.py 90000 $((3000 * K)) > $O/both_arithmetics.txt 2>&1
python scripts/upload_fuzz.py 2000 $((300 * K)) > $O/upload.txt 2>&1
python scripts/cli_fuzz.py 200 $((30 * K)) > $O/cli.txt 2>&1
python scripts/s2_fuzz.py 50000 $((3000 * K)) > $O/s2.txt 2>&1
python scripts/knob_fuzz.py 9000 $((1000 * K)) > $O/knobs.txt 2>&1
python scripts/big_fuzz.py 3000 $((150 * K)) > $O/big.txt 2>&1
python scripts/f_rows_fuzz.py 7000 $((800 * K)) > $O/f_rows.txt 2>&1
tail -n 2 $O/*.txt

"""`-m gpu`: short runs of the seeded sweeps of scripts/*_fuzz.py (the long runs are recorded in profiles/r04_fuzz_*.txt): the HIP path against the oracle on
random inputs beyond the fixed seeds of the other test files — S1 in both arithmetics, S1 -> S2, the rows after S1, large pileups, random launch knobs, the
four routes from host pileups into S1."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,first,count", [("arith_fuzz.py", 20000, 150), ("s2_fuzz.py", 20000, 80), ("f_rows_fuzz.py", 20000, 120), ("big_fuzz.py", 20000, 12),
                                                ("knob_fuzz.py", 20000, 150), ("upload_fuzz.py", 20000, 60)])
def test_seeded_sweep(script, first, count, hip_lib, oracle_mod):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), str(first), str(count)], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert "MISMATCH" not in r.stdout, r.stdout[-3000:]
    m = re.search(r"(\d+) mismatches|mismatches \{'graph': (\d+), 's2': (\d+), 'stats': (\d+), 'hapq': (\d+)\}", last)
    assert m, last
    assert all(int(x) == 0 for x in m.groups() if x is not None), last

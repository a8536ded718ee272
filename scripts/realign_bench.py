"""Dev helper: floria_hip_realign on N random windows (2 alleles each); run under rocprofv3 --kernel-trace --stats for the kernel time.
usage: scripts/realign_bench.py [N = 4000000]"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from floria_amd import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(1)
B = np.frombuffer(b"ACGT", np.uint8)
q = B[rng.integers(0, 4, size=(n, 32))]
r = q.copy()
m = rng.random((n, 32)) < 0.08
r[m] = B[rng.integers(0, 4, size=int(m.sum()))]
al = np.zeros((n, 4), np.uint8); al[:, 0] = B[rng.integers(0, 4, size=n)]; al[:, 1] = B[(np.searchsorted(B, al[:, 0]) + 1) % 4]
na = np.full(n, 2, np.uint8)
ctx = lib.FloriaHip(0)
for it in range(3):
    t = time.perf_counter(); best = ctx.realign(q, r, al, na); dt = time.perf_counter() - t
    print(f"call {it}: {n} calls in {dt * 1e3:.1f} ms through the binding (pageable H2D of {n * 69 / 1e6:.0f} MB included) = {n / dt / 1e6:.1f} M calls/s", flush=True)

"""`-m gpu`: the N > 1 leg of bench.py — process-group init, the queue broadcast through floria_amd/shard.py, the per-rank HIP contexts, the MAX / SUM
reductions and the rank-0 line — executed as TWO ranks on ONE GPU (gloo for the collectives: RCCL refuses two ranks on one device).  VERDICT r3 #4/#6: this
path had never run anywhere before the driver's 8-GPU launch; the launch line is the driver's (torch.distributed.run, 127.0.0.1)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "1", "--warmup", "0", "--contigs", "64", "--cpu-sample", "0", "--check", "0", "--pipeline", "0"]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_two_ranks_on_one_gpu_report_the_whole_job():
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *COMMON], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = last_json(one.stdout)
    env = dict(os.environ, FLORIA_BENCH_BACKEND="gloo", FLORIA_BENCH_DEVICE="0")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", *COMMON],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    d2 = last_json(two.stdout)
    assert len([l for l in two.stdout.splitlines() if l.startswith("{")]) == 1          # rank 0 prints the one line
    assert d2["n_gpus"] == 2 and d1["n_gpus"] == 1
    assert d2["scaling"] == "strong" and d2["config"]["contigs_total"] == 64
    assert d2["config"]["contigs_this_rank"] == 32                                      # the LPT queue deals equal-cost contigs evenly
    assert d2["config"]["total_blocks"] == d1["config"]["total_blocks"]                 # the SUM over ranks is the whole job
    assert d2["value"] > 0 and d2["ms_per_step"] > 0 and d2["metric"] == d1["metric"]
    # every rank's own time and share (round 5: a scaling curve is only readable with them): two entries, the blocks add up, MAX >= mean
    pr = d2["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and sum(pr["blocks"]) == d2["config"]["total_blocks"] and pr["max_over_mean"] >= 1.0
    assert abs(max(pr["ms_per_step"]) - d2["ms_per_step"]) < 0.25 * d2["ms_per_step"] and "per_rank" not in d1
    assert d2["value_is"] == "resident" and d2["value_h2d_inclusive"] > 0

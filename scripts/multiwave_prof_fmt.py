"""stdin: the stderr+stdout of one `bench.py --resident-only --steps 1 --warmup 0` run under a -DFLORIA_PROF library -> phase shares of the beam step and work per lane"""
import sys, re, json, collections
d = collections.defaultdict(float)
steps = None
for l in sys.stdin:
    if l.startswith("[prof]"):
        d.clear()
        for a, b in re.findall(r'(\d+):([0-9.]+)M', l):
            d[int(a)] = float(b) * 1e6
    elif l.startswith("{"):
        j = json.loads(l)
        steps = j["roofline"]["beam_steps_per_s"] * j["roofline"]["kernel_ms_per_step"]["beam"] * 1e-3
tot = sum(d[i] for i in range(16, 23))
print("beam phases, share of the wave cycles inside the step loop: stage %.1f%%  A (distance) %.1f%%  B (p-values, children) %.1f%%  M1 (survivors) %.1f%%  M2 (copies, zeroing) %.1f%%  adds (read-modify-write) %.1f%%  traceback %.1f%%"
      % tuple(100 * d[i] / tot for i in range(16, 23)))
print("lane-parallel phases (stage + A + M2 + adds) = %.1f%% of the chain; decision phases (B + M1: p-values, pruning, heap, survivors) = %.1f%%"
      % (100 * (d[16] + d[17] + d[20] + d[21]) / tot, 100 * (d[18] + d[19]) / tot))
if steps:
    print("per step (%.2f M steps): %.0f wave cycles; %.2f live slabs, %.1f code bytes gathered = %.1f per lane in one batch of independent loads; %.2f distinct new versions, %.1f read-modify-writes = %.1f per lane in one batch"
          % (steps / 1e6, tot / steps, d[13] / steps, d[54] / steps, d[54] / steps / 64.0, d[36] / steps, d[34] / steps, d[34] / steps / 64.0))

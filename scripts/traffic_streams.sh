#!/bin/bash
# (GPU) who asks for the bytes of a beam step: the beam kernel's REQUESTED bytes per S1 call, by stream, from the -DFLORIA_PROF counters of one resident S1 call
# of the bench workload (what the caches turn them into is the FETCH_SIZE / WRITE_SIZE side, scripts/profile_round.sh).   usage: scripts/traffic_streams.sh <tag>
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
make -C floria_amd/csrc -B EXTRA="-DFLORIA_PROF" libfloria_hip.so > /dev/null 2>&1 || { echo "BUILD FAILED"; exit 1; }
python bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --pipeline 0 --resident-only 2> gpurun_out/traffic_streams_$TAG.err > gpurun_out/traffic_streams_$TAG.out
make -C floria_amd/csrc -B libfloria_hip.so > /dev/null 2>&1
python - <<PY
import re, json, hashlib, glob
lines = [l for l in open("gpurun_out/traffic_streams_$TAG.err") if l.startswith("[prof]")]
d = {int(a): float(b) * 1e6 for a, b in re.findall(r"(\d+):([0-9.]+)M", lines[-1])}        # the last (timed) S1 call
bench = json.loads([l for l in open("gpurun_out/traffic_streams_$TAG.out") if l.startswith("{")][-1])
steps = d[9] and bench["roofline"]["beam_steps_per_s"] * bench["roofline"]["kernel_ms_per_step"]["beam"] * 1e-3
h = hashlib.sha256()
for f in sorted(glob.glob("floria_amd/csrc/*.h") + glob.glob("floria_amd/csrc/*.hip")): h.update(open(f, "rb").read())
cells, code, add_items, copy_pos, zero_items, nstates, trunc, nlive = d[9], d[54], d[34], d[32], d[35], d[15], d[37], d[13]
s = {
 "cells of the reads (LDS-DMA, 8 B per cell, read once per (block, ploidy) job)": 8 * cells,
 "per-read records and read ids (36 B per step)": 36 * steps,
 "code bytes gathered by the distance phase (1 B per (live slab, cell): one 64-B sector request each)": code,
 "read-modify-write of the sums (per (new slab version, cell): 10 B read, 5 B written)": 15 * add_items,
 "copies of a slab's live window (11 B read + 11 B written per position)": 22 * copy_pos,
 "zeroing of newly reached positions (11 B per (live slab, position))": 11 * zero_items,
 "traceback rows (4 B per survivor and step, read back once)": 8 * nstates,
 "window-exit hash terms (16 B per (live slab, leaving position) read + 16 B scratch)": 32 * trunc * (nlive / max(steps, 1)),
}
tot = sum(s.values())
out = {"tag": "$TAG", "kernel_sources_sha16": h.hexdigest()[:16], "beam_steps": steps, "requested_bytes_per_step": {k: int(v) for k, v in s.items()}, "requested_total": int(tot),
       "sector_requests_of_the_scattered_streams": {"code bytes": int(code), "read-modify-writes (3 planes each)": int(3 * add_items)},
       "note": "REQUESTED bytes of the beam kernels of one S1 call of the bench workload (-DFLORIA_PROF counters); every scattered request moves at least one 32/64-B sector, which is why the "
               "L2<->fabric counters (FETCH_SIZE / WRITE_SIZE) read far more: the code gathers and the read-modify-writes are single bytes / words in distinct lines"}
json.dump(out, open("gpurun_out/traffic_streams_$TAG.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY

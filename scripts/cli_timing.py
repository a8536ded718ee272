"""floria-hip on a metagenome-shaped synthetic data set: stage times of the batched flow against one contig per device batch.

    python scripts/cli_timing.py [--contigs 60] [--scale 0.5] [--threads 16]

Writes BAM / VCF / FASTA under $TMPDIR, runs the driver twice and prints its stage-time lines (stderr of floria-hip)."""
import argparse
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from floria_amd import synth, synth_bam  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=60)
    ap.add_argument("--scale", type=float, default=0.5)
    ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("--sub-rate", type=float, default=0.0, help="substitution errors in the reads (what realign has to absorb)")
    ap.add_argument("--arith-compare", action="store_true", help="only: the batched flow at -e 0.04 with --arith canonical against --arith reference (the default there)")
    a = ap.parse_args()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "floria_amd", "host"), "floria-hip"], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "floria_amd", "host", "floria-hip")
    tmp = tempfile.mkdtemp(prefix="floria_cli_")
    prefix = os.path.join(tmp, "d")
    t = time.time()
    cs = [synth.make_config_contig(4, i, a.scale, keep_layout=True) for i in range(a.contigs)]
    synth_bam.write_dataset(prefix, cs, seed=1, realign=False, sub_rate=a.sub_rate)
    print(f"data set: {a.contigs} contigs, {sum(c.pileup.n_reads for c in cs)} reads, {sum(len(c.snp_pos) for c in cs)} SNPs, "
          f"BAM {os.path.getsize(prefix + '.bam') >> 20} MiB, written in {time.time() - t:.1f}s; host cores {os.cpu_count()}", flush=True)
    base = [exe, "-b", prefix + ".bam", "-v", prefix + ".vcf", "-r", prefix + ".fa", "-e", "0.03125", "-l", "10000", "--snp-count-filter", "50"]
    runs = (("batched", ["-t", str(a.threads)]), ("batched, 1 thread", ["-t", "1"]), ("one contig per batch", ["-t", str(a.threads), "--batch-contigs", "1"]),
            ("ingest only: realign DP on the host", ["-t", str(a.threads), "--ingest-only"]), ("ingest only, 1 thread", ["-t", "1", "--ingest-only"]))
    if a.arith_compare:
        base[base.index("-e") + 1] = "0.04"
        runs = (("e 0.04, canonical arithmetic", ["-t", str(a.threads), "--arith", "canonical"]), ("e 0.04, reference arithmetic", ["-t", str(a.threads), "--arith", "reference"]),
                ("e 0.04, canonical arithmetic (again)", ["-t", str(a.threads), "--arith", "canonical"]), ("e 0.04, reference arithmetic (again)", ["-t", str(a.threads), "--arith", "reference"]))
    for label, extra in runs:
        out = os.path.join(tmp, "o_" + label.replace(" ", "_").replace(",", ""))
        t = time.time()
        r = subprocess.run(base + ["-o", out] + extra, capture_output=True, text=True)
        wall = time.time() - t
        if r.returncode:
            print(r.stderr[-2000:])
            raise SystemExit(1)
        lines = [ln for ln in r.stderr.splitlines() if ln.startswith(("Batches", "Total time", "Preprocessing:", "[read_bam]", "Realignment:"))]
        print(f"[{label}] wall {wall:.2f}s | " + " | ".join(lines), flush=True)


if __name__ == "__main__":
    main()

#!/bin/bash
# Pins parity: runs the REAL floria binary (bluenote-1577/floria, the commit under /root/reference) on the synthetic inputs this repository regenerates everywhere and
# lays its outputs out as tests/golden/reference_capture/ (docs/golden.md §1-§4 end to end).  Needs what this build image lacks: cargo / rustc > 1.80 (or a floria binary
# given as FLORIA_BIN), samtools (rust-htslib's IndexedReader wants a .bai / .fai), git + network for the clone (or FLORIA_SRC = a checkout).
#   usage: scripts/capture_reference.sh            -> tests/golden/reference_capture/{long,short}/e{0.03125,0.04}/...
# Afterwards:  python -m pytest tests/test_reference_capture.py -m gpu     (on a machine with an MI355X; the CPU half of the comparison - oracle vs capture - runs in
#              python -m pytest tests/test_reference_capture.py -k oracle_against)
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WORK=${WORK:-$ROOT/gpurun_out/reference_capture_work}
OUT=$ROOT/tests/golden/reference_capture
mkdir -p "$WORK" "$OUT"
cd "$WORK"
if [ -z "${FLORIA_BIN:-}" ]; then
  SRC=${FLORIA_SRC:-$WORK/floria}
  [ -d "$SRC" ] || git clone https://github.com/bluenote-1577/floria "$SRC"
  (cd "$SRC" && cargo build --release)
  FLORIA_BIN=$SRC/target/release/floria
  (cd "$SRC" && { rustc --version; cargo --version; git rev-parse HEAD 2>/dev/null || true; grep -A1 'name = "hashbrown"' Cargo.lock || true; }) > "$OUT/rustc_version.txt"
else
  { echo "binary given as FLORIA_BIN=$FLORIA_BIN"; rustc --version 2>/dev/null || echo "rustc: unknown"; } > "$OUT/rustc_version.txt"
fi
# ---- inputs: the same seeded data sets tests/test_reference_capture.py regenerates (no htslib needed to WRITE them: floria_amd/synth_bam.py)
PYTHONPATH=$ROOT python - <<'PY'
from floria_amd import synth, synth_bam
# long reads: the quick-start substitute (config 1) and a half-size config-4 contig; -l 10000
synth_bam.write_dataset("golden_long", [synth.make_config_contig(1, 0, keep_layout=True), synth.make_config_contig(4, 3, 0.5, keep_layout=True)], seed=7)
# paired short reads (a BASELINE config-3 contig at 0.3 scale): the case where the S2 visiting order, a14's dropped read and the merged fragments' set orders matter; -l 500
synth_bam.write_dataset("golden_short", [synth.make_config_contig(3, 2, 0.3, keep_layout=True)], seed=7)
PY
for D in long short; do samtools index golden_$D.bam; samtools faidx golden_$D.fa; done
# ---- runs: single-threaded with the trace dumps (MEC vector per block: graph_processing.rs:258-266; local_parts/: :289-300).  0.03125 is dyadic: every f64 sum is exact,
# both arithmetics coincide, S1 does not depend on hash orders - any mismatch there is a bug.  0.04 is the realistic regime: compared with the oracle's arithmetic mode 1
# and with the product's `arith = 1` (what floria-hip runs at that epsilon).
for D in long short; do
  L=10000; [ $D = short ] && L=500
  for E in 0.03125 0.04; do
    rm -rf out_${D}_e$E
    "$FLORIA_BIN" -b golden_$D.bam -v golden_$D.vcf -r golden_$D.fa -o out_${D}_e$E -e $E -l $L -t 1 --trace > trace_${D}_e$E.log 2>&1 ||        # (simple_logger's stream differs between its versions: both) { echo "floria failed on $D at -e $E: see $WORK/trace_${D}_e$E.log"; exit 1; }
    T=$OUT/$D/e$E
    rm -rf "$T"; mkdir -p "$T"
    grep "MEC vector" trace_${D}_e$E.log > "$T/trace.log" || true
    cp out_${D}_e$E/contig_ploidy_info.tsv "$T/"
    for C in out_${D}_e$E/*/; do
      N=$(basename "$C")
      [ -f "$C/$N.haplosets" ] || continue
      mkdir -p "$T/$N"
      cp "$C/$N.haplosets" "$C/$N.vartigs" "$C/vartig_info.txt" "$T/$N/"
      [ -d "$C/local_parts" ] && cp -r "$C/local_parts" "$T/$N/"
    done
  done
done
echo "captured: $(find "$OUT" -type f | wc -l) files under $OUT — commit them (data, not reference source) and run tests/test_reference_capture.py"

// floria-hip — command-line driver with the reference's flags (floria.rs:21-196, parse_cmd_line.rs:11-196) and its per-contig
// flow (floria.rs:229-388): ingest -> generate_hap_graph (device) -> solve_lp_graph -> get_disjoint_paths_rewrite ->
// process_reads_for_final_parts (device) -> get_frags_in_snpless_gaps -> write_outputs.  The contigs take that flow in BATCHES:
// the host stages of a batch run one contig per task on -t threads, every device stage runs once per batch (a metagenome is
// thousands of small contigs; one device call per contig and stage would be launch-latency-bound).  Files are those of the
// per-contig flow, byte for byte.
//
//   floria-hip -b reads.bam -v calls.vcf -r reference.fa -o results [-e 0.04] [-l 10000] [-n 10] [-p 5] [-d 0.0005] [-s 2] [-m 15]
//              [-t 10] [-G contig ...] [-X] [--no-stop-heuristic] [--snp-count-filter 100] [--supp-aln-dist-cutoff 40000] [--overwrite]
//
//              [--output-reads [--gzip-reads] [--extra-trimming]] [--ignore-monomorphic]
// Flags of the reference that are not supported and say so: the hidden -H/--hybrid, --reassign-short, --bin-by-cov (-q is accepted and
// ignored, as in the reference).  Extras: --device N, --devices LIST (e.g. 0-7 or 0,2,5: the contigs of a batch are dealt to these GPUs of the node,
// longest first; one context and one host thread per device; the files are those of a one-GPU run),
// --batch-contigs N / --batch-cells N (size of a device batch), --debug (debug_graph.txt per contig), --lp-tie first|last (which optimal vertex of the
// stitching LP is used where the optimum is not unique; the run reports how many contigs that concerns), and for tests --dump-frags FILE,
// --ingest-only, --no-realign, --stitch-graph FILE.
#include <sys/stat.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

#include "floria_host.hpp"

using namespace floria;

namespace {

void usage() {
    fputs("floria-hip - strain phasing for short or long-read shotgun metagenomic sequencing (MI355X build of floria's phasing path).\n\n"
          "Example usage :\nfloria-hip -b bamfile.bam -v vcffile.vcf -r reference.fa -o results -t 10\n\n"
          "floria's options, same meaning:\n"
          "  -b FILE  sorted BAM            -v FILE  VCF (plain / gz / bgz) or BCF with the SNPs     -r FILE  reference fasta\n"
          "  -o DIR   output directory [floria_out_dir]       --overwrite   reuse an existing one    -t N  host threads [10]\n"
          "  -e X     allele error rate epsilon [estimated]   -l N  block length in bases [estimated]   -d X  SNP density filter [0.0005]\n"
          "  -n N     beam solutions [10]    -p N  maximum ploidy [5; this build: 1..16]    -s 1|2|3  ploidy sensitivity [2]    --no-stop-heuristic\n"
          "  -m N     MAPQ cutoff [15]       -G / --contigs NAME..  only these contigs      --snp-count-filter N [100]\n"
          "  -X / --no-supp  ignore supplementary alignments      --supp-aln-dist-cutoff N [40000]      --ignore-monomorphic\n"
          "  --output-reads [--gzip-reads] [--extra-trimming]     --debug / --trace\n"
          "this build's own:\n"
          "  --device N | --devices 0-7 | 0,2,5   GPU(s); contigs of a batch are dealt to them, files do not depend on it\n"
          "  --batch-contigs N [4096]  --batch-cells N [2^28]   contigs taken through the device stages together\n"
          "  --bam-window-mb N [512]   inflated BAM records held in memory at a time\n"
          "  --arith auto|reference|canonical   how sums of epsilon terms are rounded when -e is not a multiple of 2^-10: 'reference' = floria's running f64 sums\n"
          "                 in its hash containers' iteration order (as far as that order can be emulated), 'canonical' = every sum rounded once; auto [default] =\n"
          "                 reference, except for batches with fragments merged from mates / supplementary alignments or under --ignore-monomorphic\n"
          "  --epsilon-round   round an ESTIMATED -e to a multiple of 2^-10 (there both arithmetics are the same function)\n"
          "  --lp-tie first|last, --lp-report   which optimal vertex of the stitching LP is used where the optimum is not unique, and how often that is\n"
          "  --no-realign      skip the re-alignment of the reads' bases around SNPs\n"
          "where this build is not floria (DESIGN.md sections 6 and 7):\n"
          "  * re-alignment around SNPs is the EXACT affine-gap alignment of the 32-base windows; floria's block-aligner is a banded heuristic of it (calls can differ)\n"
          "  * the stitching LP is solved exactly as a min-cost flow: the same optimal value as floria's simplex, possibly another optimal vertex (short reads: haplosets can differ)\n"
          "  * hash-set iteration orders decide ties in floria (visiting order of the final reassignment, the read dropped at a haplogroup split): here ascending read order\n"
          "  * not supported: -H / --hybrid, --reassign-short, --bin-by-cov (hidden or beta in floria)\n", stderr);
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

// one task per index on up to `threads` std::threads; the first exception is rethrown on the caller
template <class F> static void parallel_for(size_t n, size_t threads, F&& f) {
    threads = std::min(threads, n);
    if (threads <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
    std::atomic<size_t> next{0};
    std::exception_ptr err;
    std::mutex mu;
    std::vector<std::thread> pool;
    for (size_t t = 0; t < threads; ++t)
        pool.emplace_back([&] {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n) return;
                try { f(i); } catch (...) { std::lock_guard<std::mutex> g(mu); if (!err) err = std::current_exception(); next = n; return; }
            }
        });
    for (std::thread& t : pool) t.join();
    if (err) std::rethrow_exception(err);
}

// hap graph, flows and paths of a contig in the N / E / F / P line format (--debug; tests/test_gpu_cli.py reads it back)
static void write_debug_graph(const ContigWork& w) {
    struct stat st;
    if (stat(w.out_dir.c_str(), &st) != 0) mkdir(w.out_dir.c_str(), 0777);
    std::ofstream g(w.out_dir + "/debug_graph.txt", std::ios::trunc);
    g.precision(17);
    for (const auto& col : w.hap_graph) for (const HapNode& n : col) {
        g << "N\t" << n.column << "\t" << n.row << "\t" << n.id << "\t" << n.cov << "\t" << n.snp_endpoints.first << "\t" << n.snp_endpoints.second;
        for (const Frag* f : n.frag_set) g << "\t" << f->counter_id;
        g << "\n";
    }
    for (const auto& col : w.hap_graph) for (const HapNode& n : col) for (const auto& e : n.out_edges) g << "E\t" << n.column << "\t" << n.row << "\t" << e.first << "\t" << e.second << "\n";
    for (const FlowUpdate& f : w.flows) g << "F\t" << f.n1.first << "\t" << f.n1.second << "\t" << f.n2.second << "\t" << f.flow << "\n";
    for (size_t k = 0; k < w.path_parts.size(); ++k) {
        g << "P\t" << w.path_ranges[k].first << "\t" << w.path_ranges[k].second;
        for (const Frag* f : w.path_parts[k]) g << "\t" << f->counter_id;
        g << "\n";
    }
}

int main(int argc, char** argv) {
    floria_hip_init_env();                      // GPU_MAX_HW_QUEUES=12 unless set: the job groups' streams and the copy streams must not share hardware queues (read when HIP initialises)
    Options o;
    bool have_e = false, have_l = false;
    std::string dump_frags;
    size_t batch_contigs = 4096;         // --batch-contigs
    uint64_t batch_cells = 256ull << 20; // --batch-cells (1.5 GB of pinned staging)
    bool ingest_only = false, no_realign = false;
    std::string stitch_graph;
    LpTie lp_tie = LpTie::First;         // --lp-tie first|last: which optimal vertex of the stitching LP is used where the optimum is not unique (stitch.cpp)
    std::atomic<size_t> lp_contigs{0}, lp_not_unique{0}, lp_edges{0}, lp_movable{0};
    bool lp_report = false;              // --lp-report (implied by --debug): count the edges whose flow differs in another optimal solution of the stitching LP
    bool debug = false;                  // --debug / --trace: per contig, debug_graph.txt (hap graph, LP flows, joined paths) next to the outputs
    std::vector<int> devices;            // --devices: the GPUs the contigs of a batch are dealt to (empty: --device alone)
    size_t bam_window = (size_t)512 << 20;   // --bam-window-mb: inflated BAM bytes held at a time (more only when one contig alone is larger)
    std::string arith_opt = "auto";      // --arith auto|reference|canonical: the reference's running f64 sums (device mode arith = 1) or the exact (Q24, #eps) form; auto = reference
                                         // arithmetic exactly when epsilon is not a multiple of 2^-10 (where the two are different functions)
    bool eps_round = false;              // --epsilon-round: round an auto-estimated -e to a multiple of 2^-10 (default: used as estimated, like the reference)
    std::string run_note;                // second line of cmd.log
    std::thread freer;                   // frees the Frags of the previous batch while the next one is ingested; joined before the process leaves main
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } freer_guard{freer};
    try {
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            auto val = [&]() -> std::string { if (i + 1 >= argc) throw Error(FLORIA_E_INVALID, "missing value for " + a); return argv[++i]; };
            if (a == "-b") o.bam_file = val();
            else if (a == "-v") o.vcf_file = val();
            else if (a == "-r") o.reference_fasta = val();
            else if (a == "-o" || a == "--output-dir") o.out_dir = val();
            else if (a == "-t" || a == "--threads") o.num_threads = (size_t)std::stoul(val());
            else if (a == "-e" || a == "--epsilon") { o.epsilon = std::stod(val()); have_e = true; }
            else if (a == "-n" || a == "--beam-solns") o.max_number_solns = (size_t)std::stoul(val());
            else if (a == "-p" || a == "--max-ploidy") {
                o.max_ploidy = (size_t)std::stoul(val());
                if (o.max_ploidy < 1 || o.max_ploidy > FLORIA_MAX_PLOIDY)           // (the reference takes any -p; its default is 5.  include/floria_hip.h: the optimise kernel packs partitions into 4 bits)
                    throw Error(FLORIA_E_UNSUPPORTED, "-p " + std::to_string(o.max_ploidy) + ": this build phases ploidies 1 to " + std::to_string(FLORIA_MAX_PLOIDY));
            }
            else if (a == "-l" || a == "--block-length") { o.block_length = (size_t)std::stoul(val()); have_l = true; }
            else if (a == "-d" || a == "--snp-density") o.snp_density = std::stod(val());
            else if (a == "-s" || a == "--ploidy-sensitivity") o.ploidy_sensitivity = (uint8_t)std::stoul(val());
            else if (a == "-m" || a == "--mapq-cutoff") o.mapq_cutoff = (uint8_t)std::stoul(val());
            else if (a == "--snp-count-filter") o.snp_count_filter = (size_t)std::stoul(val());
            else if (a == "--supp-aln-dist-cutoff") o.supp_aln_dist_cutoff = std::stoll(val());
            else if (a == "-X" || a == "--no-supp") o.dont_use_supp_aln = true;
            else if (a == "--no-stop-heuristic") o.stopping_heuristic = false;
            else if (a == "--overwrite") o.overwrite = true;
            else if (a == "--debug" || a == "--trace") { debug = true; lp_report = true; }
            else if (a == "--lp-report") lp_report = true;
            else if (a == "--lp-tie") { const std::string v = val(); if (v != "first" && v != "last") throw Error(FLORIA_E_INVALID, "--lp-tie takes first or last"); lp_tie = v == "last" ? LpTie::Last : LpTie::First; }
            else if (a == "-q") {}
            else if (a == "-G" || a == "--contigs") { while (i + 1 < argc && argv[i + 1][0] != '-') o.list_to_phase.push_back(argv[++i]); }
            else if (a == "--device") o.device = std::stoi(val());
            else if (a == "--epsilon-as-estimated") {}                          // (the default since round 4; accepted for older command lines)
            else if (a == "--epsilon-round") eps_round = true;
            else if (a == "--arith") { arith_opt = val(); if (arith_opt != "auto" && arith_opt != "reference" && arith_opt != "canonical") throw Error(FLORIA_E_INVALID, "--arith takes auto, reference or canonical"); }
            else if (a == "--bam-window-mb") bam_window = std::max<size_t>(1, std::stoul(val())) << 20;
            else if (a == "--bam-window-kb") bam_window = std::max<size_t>(1, std::stoul(val())) << 10;       // (tests: many segments on a small file)
            else if (a == "--devices") {                                 // "0-7", "0,2,5", "0-3,6"; a device may be named twice (two contexts on it)
                std::stringstream ls(val());
                std::string item;
                while (std::getline(ls, item, ',')) {
                    const size_t dash = item.find('-', 1);
                    const int lo = std::stoi(item.substr(0, dash)), hi = dash == std::string::npos ? lo : std::stoi(item.substr(dash + 1));
                    if (lo < 0 || hi < lo || hi - lo > 1024) throw Error(FLORIA_E_INVALID, "--devices: bad range " + item);
                    for (int d = lo; d <= hi; ++d) devices.push_back(d);
                }
                if (devices.empty()) throw Error(FLORIA_E_INVALID, "--devices: empty list");
            }
            else if (a == "--vcf-profile") {                             // (tests) FILE contig ... -> "contig n_snps" and one "pos alleles" line per SNP, no GPU needed
                const std::string vcf = val();
                std::vector<std::string> names;
                while (i + 1 < argc) names.push_back(argv[++i]);
                const VcfProfile vp = get_vcf_profile(vcf, names);
                for (const auto& kv : vp.snp_to_genome_pos) {
                    printf("#%s\t%zu\n", kv.first.c_str(), kv.second.size());
                    const auto& pam = vp.vcf_pos_allele_map.at(kv.first);
                    for (GnPosition g : kv.second) { printf("%zu\t", (size_t)g); for (Genotype x : pam.at(g)) putchar((char)x); putchar('\n'); }
                }
                return 0;
            }
            else if (a == "--lpt-assign") {                              // (tests) world cost cost ... -> the device of every item, no GPU needed
                const uint32_t world = (uint32_t)std::stoul(val());
                std::vector<double> costs;
                while (i + 1 < argc) costs.push_back(std::stod(argv[++i]));
                for (uint32_t d : lpt_assign(costs, world)) printf("%u\n", d);
                return 0;
            }
            else if (a == "--batch-contigs") batch_contigs = std::max<size_t>(1, std::stoul(val()));       // contigs per device batch
            else if (a == "--batch-cells") batch_cells = std::max<uint64_t>(1, std::stoull(val()));        // SNP calls per device batch
            else if (a == "--dump-frags") dump_frags = val();
            else if (a == "--no-realign") no_realign = true;                   // (tests) keep the alleles as called
            else if (a == "--ingest-only") ingest_only = true;          // (tests) stop after ingest: needs no GPU
            else if (a == "--stitch-graph") stitch_graph = val();       // (tests) N / E lines of a hap graph -> F / P lines on stdout: needs no GPU
            else if (a == "-h" || a == "--help") { usage(); return 0; }
            else if (a == "--output-reads") o.output_reads = true;
            else if (a == "--gzip-reads") o.gzip = true;
            else if (a == "--extra-trimming") o.trim_reads = true;
            else if (a == "--ignore-monomorphic") o.ignore_monomorphic = true;
            else if (a == "-H" || a == "--hybrid" || a == "--reassign-short" || a == "--bin-by-cov")
                throw Error(FLORIA_E_UNSUPPORTED, "option " + a + " of floria is not supported by floria-hip");
            else throw Error(FLORIA_E_INVALID, "unknown option " + a);
        }
        if (!stitch_graph.empty()) {
            // hap graph in the format of debug_graph.txt (N: column row id cov lo hi reads..., E: column row row2 weight)
            std::ifstream in(stitch_graph);
            if (!in) throw Error(FLORIA_E_INVALID, "cannot open " + stitch_graph);
            std::vector<std::vector<HapNode>> hg;
            std::vector<Frag> frags;
            std::vector<std::vector<std::vector<size_t>>> node_reads;
            std::string tag;
            size_t max_read = 0;
            struct EdgeIn { size_t c, r, r2; double w; };
            std::vector<EdgeIn> edges;
            std::string line;
            while (std::getline(in, line)) {
                std::istringstream ls(line);
                ls >> tag;
                if (tag == "N") {
                    size_t c, r, id; double cov; SnpPosition lo, hi;
                    ls >> c >> r >> id >> cov >> lo >> hi;
                    if (hg.size() <= c) { hg.resize(c + 1); node_reads.resize(c + 1); }
                    if (hg[c].size() <= r) { hg[c].resize(r + 1); node_reads[c].resize(r + 1); }
                    HapNode& n = hg[c][r];
                    n.column = c; n.row = r; n.id = id; n.cov = cov; n.snp_endpoints = {lo, hi};
                    size_t x;
                    while (ls >> x) { node_reads[c][r].push_back(x); max_read = std::max(max_read, x + 1); }
                } else if (tag == "E") { EdgeIn e; ls >> e.c >> e.r >> e.r2 >> e.w; edges.push_back(e); }
            }
            frags.resize(max_read);
            for (size_t i = 0; i < max_read; ++i) frags[i].counter_id = i;
            for (size_t c = 0; c < hg.size(); ++c) for (size_t r = 0; r < hg[c].size(); ++r) for (size_t x : node_reads[c][r]) hg[c][r].frag_set.push_back(&frags[x]);
            for (const EdgeIn& e : edges) { hg[e.c][e.r].out_edges.push_back({e.r2, e.w}); hg[e.c + 1][e.r2].in_edges.push_back({e.r, e.w}); }
            LpInfo li;
            const FlowUpVec fl = solve_lp_graph(hg, lp_tie, &li);
            auto paths = get_disjoint_paths_rewrite(hg, fl, o);
            printf("U\t%lld\t%zu\n", (long long)li.cost, li.movable_edges);
            for (const FlowUpdate& f : fl) printf("F\t%zu\t%zu\t%zu\t%.17g\n", f.n1.first, f.n1.second, f.n2.second, f.flow);
            for (size_t k = 0; k < paths.first.size(); ++k) {
                printf("P\t%u\t%u", paths.second[k].first, paths.second[k].second);
                for (const Frag* f : paths.first[k]) printf("\t%zu", f->counter_id);
                printf("\n");
            }
            return 0;
        }
        if (o.bam_file.empty()) throw Error(FLORIA_E_INVALID, "Must input a BAM file.");
        if (o.vcf_file.empty() || o.reference_fasta.empty()) throw Error(FLORIA_E_INVALID, "-v and -r are required");
        if (!(o.ploidy_sensitivity >= 1 && o.ploidy_sensitivity <= 3)) throw Error(FLORIA_E_INVALID, "Ploidy sensitivty option must be between 1 and 3");

        const double t_all = now_s();
        fprintf(stderr, "Preprocessing VCF/Reference\n");
        double tp = now_s();
        // the BAM is streamed: the records of a run of complete contigs at a time (about --bam-window-mb of inflated records), never the whole file
        BamStream stream(o.bam_file, std::max<size_t>(1, o.num_threads));
        double t_bam = now_s() - tp;
        // ---- the epsilon policy (DESIGN.md "Arithmetic").  Every weighted sum of the phasing path is an exact multiple of 2^-24; the only inexact
        // terms of the reference are its running `+= epsilon` additions, whose rounding depends on hash-map iteration order unless epsilon is dyadic.
        // For an epsilon that is a multiple of 2^-10 every one of those sums is exact in f64 in ANY order, so the function this library computes IS the
        // reference's function.  An auto-estimated epsilon (parse_cmd_line.rs:72-90) is used AS ESTIMATED, like the reference does, and cmd.log has the
        // reference's single line; --epsilon-round rounds it to the nearest multiple of 2^-10 (a change of at most 0.0005 to a coverage-sampled average)
        // and records both values on a second line of cmd.log.  Whatever its origin, an epsilon that is not such a multiple gets one warning on stderr:
        // results are then an equally good solution, but may differ from the Rust binary's where scores tie in exact arithmetic.
        auto dyadic10 = [](double e) { const double k = e * 1024.0; return k == std::floor(k); };
        if (!have_e || !have_l) {                                                     // parse_cmd_line.rs:72-90
            tp = now_s();
            EpsilonEstimator estimator;                                               // (a first pass over as much of the file as 1000 sampled columns need)
            { BamFile seg; while (!estimator.done() && stream.next(seg, bam_window)) estimator.feed(seg); }
            stream.rewind();
            const auto est = estimator.result();
            t_bam += now_s() - tp;
            if (!have_l) o.block_length = est.first;
            if (!have_e) {
                o.epsilon = est.second;
                if (eps_round) {
                    o.epsilon = std::max(1.0, std::floor(est.second * 1024.0 + 0.5)) / 1024.0;
                    char buf[256];
                    snprintf(buf, sizeof buf, "# floria-hip: -e estimated %.17g, used %.10g (rounded to a multiple of 2^-10: every sum of the phasing path is then exact in f64; "
                                              "requested by --epsilon-round)", est.second, o.epsilon);
                    run_note = buf;
                }
            }
            fprintf(stderr, "Estimated -l %zu, -e %g (used where not given: -l %zu, -e %.10g)\n", est.first, est.second, o.block_length, o.epsilon);
        }
        const bool reference_arith = arith_opt == "reference" || (arith_opt == "auto" && !dyadic10(o.epsilon));
        if (!dyadic10(o.epsilon)) {
            const double near = std::max(1.0, std::floor(o.epsilon * 1024.0 + 0.5)) / 1024.0;
            if (reference_arith)
                fprintf(stderr, "floria-hip: note: -e %.17g is not a multiple of 2^-10: phasing in floria's own running-sum arithmetic (sums of epsilon terms rounded term by term, in the "
                                "iteration order of its hash containers; slower kernels).  --arith canonical keeps the fast kernels (every sum rounded once), -e %.10g is an epsilon at which "
                                "the two are the same function\n", o.epsilon, near);
            else
                fprintf(stderr, "floria-hip: warning: -e %.17g is not a multiple of 2^-10; with --arith canonical sums of epsilon terms are rounded once here and term by term (in hash-map order) in floria, "
                                "so haplosets can differ from floria's in exact ties (-e %.10g, or --epsilon-round for an estimated one, avoids that)\n", o.epsilon, near);
        }
        if (!ingest_only) write_run_files(o, argc, argv, run_note);
        const std::vector<std::string> contigs = stream.target_names();                 // get_contigs_to_phase (file_reader.rs:738-746)
        tp = now_s();
        const VcfProfile vp = get_vcf_profile(o.vcf_file, contigs);
        const std::map<std::string, std::string> fasta = get_fasta_seqs(o.reference_fasta);
        const double t_vcf = now_s() - tp;
        tp = now_s();
        // one context per device of --devices (default: the one --device); every one throws without a usable MI355X: no CPU fallback
        if (devices.empty()) devices.push_back(o.device);
        std::vector<std::unique_ptr<Session>> sessions;
        if (!ingest_only) for (int d : devices) sessions.emplace_back(new Session(d));
        for (auto& s : sessions) if (floria_hip_set_option(s->ctx(), "arith", reference_arith ? 1 : 0) != 0) throw Error(FLORIA_E_INVALID, floria_hip_last_error());
        // The reference-arithmetic mode adds a read's terms in the iteration order of its position set.  For a fragment built from ONE alignment the library emulates that
        // set on the device (arith_kernel.h); fragments whose set was EXTENDED by a mate or a supplementary piece (file_reader.rs:541, 639) or cut down by
        // --ignore-monomorphic (utils_frags.rs:745-755) carry what their set went through (Frag::position_segments / removed_positions, ingest.cpp) and a Batch hands
        // the replayed order to the library (floria_pileup_packed::set_order): round 6; until then such batches fell back to the canonical form under --arith auto.
        size_t n_batches_orders = 0, n_frags_merged = 0, n_frags_cut = 0;
        Session* const session_holder = sessions.empty() ? nullptr : sessions[0].get();
        fprintf(stderr, "Preprocessing: BAM header%s %.3fs, VCF + FASTA %.3fs, device %.3fs\n", (!have_e || !have_l) ? " + parameter estimate" : "", t_bam, t_vcf, now_s() - tp);

        std::ofstream dump;
        if (!dump_frags.empty()) dump.open(dump_frags, std::ios::trunc);
        // ---- which contigs (floria.rs:229-262) -------------------------------------------------------------------------------------
        bool warn_first_length = true;
        const size_t n_threads = std::max<size_t>(1, o.num_threads);
        double t_ingest = 0., t_s1 = 0., t_stitch = 0., t_s2 = 0., t_stats = 0., t_write = 0., t_realign = 0., t_stream = 0.;
        size_t n_realign_device = 0, n_batches = 0, n_records = 0, n_segments = 0;
        // every contig that will be phased must be in the reference FASTA: said before anything is written, not when its batch comes up
        if (!ingest_only)
            for (const std::string& contig : contigs) {
                if (!o.list_to_phase.empty() && std::find(o.list_to_phase.begin(), o.list_to_phase.end(), contig) == o.list_to_phase.end()) continue;
                const auto pam = vp.vcf_pos_allele_map.find(contig);
                if (pam == vp.vcf_pos_allele_map.end() || pam->second.size() < o.snp_count_filter) continue;
                if (fasta.find(contig) == fasta.end()) throw Error(FLORIA_E_INVALID, "contig " + contig + " is not in the reference fasta");
            }
        BamFile bam;                                  // the current segment: every record of the contigs [tid_begin, tid_end)
        for (;;) {
        { const double ts = now_s(); const bool more = stream.next(bam, bam_window); t_stream += now_s() - ts; if (!more) break; }
        ++n_segments; n_records += bam.records.size();
        std::vector<std::string> todo;
        for (int32_t tid = bam.tid_begin; tid < bam.tid_end; ++tid) {
            const std::string& contig = contigs[(size_t)tid];
            if (!o.list_to_phase.empty() && std::find(o.list_to_phase.begin(), o.list_to_phase.end(), contig) == o.list_to_phase.end()) continue;
            const auto pam = vp.vcf_pos_allele_map.find(contig);
            if (pam == vp.vcf_pos_allele_map.end() || pam->second.size() < o.snp_count_filter) {
                if (warn_first_length)
                    fprintf(stderr, "A contig (%s) is not present or has < %zu variants. This warning will not be shown from now on. Make sure to change --snp-count-filter if you want to phase small contigs.\n",
                            contig.c_str(), o.snp_count_filter);
                warn_first_length = false;
                continue;
            }
            todo.push_back(contig);
        }
        // The contigs go through the stages in batches (floria_host.hpp, "many contigs at once"): the host stages of a batch run on -t
        // threads, one contig per task; the device stages once per batch.  A batch is closed at batch_contigs contigs or batch_cells
        // SNP calls, whichever comes first (the pinned staging buffer holds 6 bytes per call).
        size_t done = 0;
        while (done < todo.size()) {
            // ---- ingest (floria.rs:264-293), one contig per task ----------------------------------------------------------------------
            double t0 = now_s();
            std::vector<ContigWork> work;
            uint64_t cells = 0;
            while (done < todo.size() && work.size() < batch_contigs && cells < batch_cells) {
                const size_t take = std::min({todo.size() - done, batch_contigs - work.size(), std::max<size_t>(n_threads * 2, 2)});
                std::vector<ContigWork> got(take);
                // records -> Frags with the realignment's undecided windows queued (one queue per contig), the queued windows of the
                // whole round scored by ONE device call, then the merge of mates / supplementary pieces and the sort
                std::vector<std::unique_ptr<ContigIngest>> ing(take);
                std::vector<RealignQueue> queues(take);
                const bool on_device = !ingest_only && !no_realign;
                parallel_for(take, n_threads, [&](size_t i) {
                    const std::string& contig = todo[done + i];
                    const auto fa = fasta.find(contig);
                    ing[i].reset(new ContigIngest(bam, vp, o, contig, (fa != fasta.end() && !no_realign) ? &fa->second : nullptr, on_device ? &queues[i] : nullptr));
                });
                if (on_device) {
                    const double tr = now_s();
                    RealignQueue all;
                    for (RealignQueue& q : queues) all.append(std::move(q));
                    n_realign_device += all.size();
                    realign_queue_on_device(*session_holder, all);
                    t_realign += now_s() - tr;
                }
                parallel_for(take, n_threads, [&](size_t i) {
                    const std::string& contig = todo[done + i];
                    ContigWork& w = got[i];
                    w.name = contig; w.out_dir = o.out_dir + "/" + contig;
                    const auto fa = fasta.find(contig);
                    auto fr = ing[i]->finish();
                    ing[i].reset();
                    w.all_frags = std::move(fr.first); w.frags_without_snps = std::move(fr.second);
                    const auto sgp = vp.snp_to_genome_pos.find(contig);
                    w.snp_to_genome_pos = sgp == vp.snp_to_genome_pos.end() ? nullptr : &sgp->second;
                    // the contig directory is (re)made only for a contig that has fragments and SNPs, as floria.rs:263-281 does: with --overwrite a contig
                    // that yields nothing in this run keeps what an earlier run wrote
                    if (!ingest_only && !w.all_frags.empty() && w.snp_to_genome_pos) prepare_contig_dir(w.out_dir, o);
                    w.contig_len = fa == fasta.end() ? 0 : fa->second.size();
                    std::sort(w.all_frags.begin(), w.all_frags.end());                     // floria.rs:289-293
                    for (size_t k = 0; k < w.all_frags.size(); ++k) w.all_frags[k].counter_id = k;
                    if (o.ignore_monomorphic) w.all_frags = remove_monomorphic_allele(std::move(w.all_frags), o.epsilon);     // floria.rs:315-317
                });
                done += take;
                for (ContigWork& w : got) {
                    fprintf(stderr, "Number of reads passing filtering: %zu (%s)\n", w.all_frags.size(), w.name.c_str());
                    if (w.all_frags.empty() || !w.snp_to_genome_pos) continue;
                    if (!ingest_only && w.contig_len == 0) throw Error(FLORIA_E_INVALID, "contig " + w.name + " is not in the reference fasta");
                    for (const Frag& f : w.all_frags) cells += f.seq_dict.size();
                    work.push_back(std::move(w));
                }
            }
            if (dump.is_open())
                for (const ContigWork& w : work) {
                    dump << "#CONTIG\t" << w.name << "\t" << w.all_frags.size() << "\t" << w.frags_without_snps.size() << "\n";
                    for (const Frag& f : w.all_frags) {
                        dump << f.id << "\t" << f.first_position << "\t" << f.last_position << "\t" << f.first_pos_base << "\t" << f.last_pos_base << "\t" << (f.is_paired ? 1 : 0);
                        for (const auto& kv : f.seq_dict) dump << "\t" << kv.first << ":" << (int)kv.second << ":" << (int)f.qual_dict.at(kv.first);
                        dump << "\n";
                        if (f.other_set_order()) {          // the iteration order of its position set, replayed (tests compare it with the oracle's emulation)
                            dump << "#ORDER";
                            for (SnpPosition sp : f.positions_order()) dump << "\t" << sp;
                            dump << "\n";
                        }
                    }
                    for (const Frag& f : w.frags_without_snps) dump << "#SNPLESS\t" << f.id << "\t" << f.first_pos_base << "\t" << f.last_pos_base << "\t" << (f.seq_len[0] + f.seq_len[1]) << "\n";
                }
            t_ingest += now_s() - t0;
            if (ingest_only || work.empty()) continue;
            ++n_batches;
            if (reference_arith) {
                size_t merged = 0, cut = 0;
                for (const ContigWork& w : work) for (const Frag& f : w.all_frags) { merged += f.merged_positions ? 1 : 0; cut += f.removed_positions.empty() ? 0 : 1; }
                n_frags_merged += merged; n_frags_cut += cut;
                if (merged || cut) ++n_batches_orders;
            }
            // the device stages of a set of contigs on one context: S1 + hap graph in one pipelined call, LP + path peeling on the host (one contig per
            // task), S2, COV / ERR / HAPQ of the final haplosets.  `tm` receives the wall seconds of the four stages.
            auto device_stages = [&](Session& session, std::vector<ContigWork>& part, size_t threads, double* tm) {
                double t1 = now_s();
                Batch batch(session, part, reference_arith);
                batch.generate_hap_graphs(o);
                tm[0] += now_s() - t1; t1 = now_s();
                parallel_for(part.size(), threads, [&](size_t i) {
                    ContigWork& w = part[i];
                    // (the uniqueness diagnostics cost O(E (V + E)) on top of the solve — 60-70 % more on layered graphs of thousands of columns — and feed
                    // one summary line: only under --debug / --lp-report)
                    LpInfo li;
                    w.flows = solve_lp_graph(w.hap_graph, lp_tie, lp_report ? &li : nullptr);
                    if (lp_report) { ++lp_contigs; lp_not_unique += li.movable_edges != 0; lp_edges += w.flows.size(); lp_movable += li.movable_edges; }
                    auto paths = get_disjoint_paths_rewrite(w.hap_graph, w.flows, o);
                    w.path_parts = std::move(paths.first); w.path_ranges = std::move(paths.second);
                    if (debug) write_debug_graph(w);
                });
                tm[1] += now_s() - t1; t1 = now_s();
                batch.process_reads_for_final_parts(o);
                tm[2] += now_s() - t1; t1 = now_s();
                batch.stats_and_hapq(o);
                tm[3] += now_s() - t1;
            };
            double tm[4] = {0., 0., 0., 0.};
            if (sessions.size() == 1) device_stages(*sessions[0], work, n_threads, tm);
            else {
                // ---- the contigs of the batch dealt to the devices (longest first, by SNP calls), one host thread per device; every contig comes back
                // to its place, so what follows (writers, the contig table) sees the batch in contig order whatever the dealing was
                std::vector<double> cost(work.size());
                for (size_t i = 0; i < work.size(); ++i) { double c = 0; for (const Frag& f : work[i].all_frags) c += (double)f.seq_dict.size(); cost[i] = c; }
                const std::vector<uint32_t> owner = lpt_assign(cost, (uint32_t)sessions.size());
                std::vector<std::vector<ContigWork>> part(sessions.size());
                std::vector<std::vector<size_t>> origin(sessions.size());
                for (size_t i = 0; i < work.size(); ++i) { part[owner[i]].push_back(std::move(work[i])); origin[owner[i]].push_back(i); }
                std::vector<std::array<double, 4>> tms(sessions.size(), std::array<double, 4>{0., 0., 0., 0.});
                const size_t per_dev = std::max<size_t>(1, n_threads / sessions.size());
                parallel_for(sessions.size(), sessions.size(), [&](size_t d) { if (!part[d].empty()) device_stages(*sessions[d], part[d], per_dev, tms[d].data()); });
                for (size_t d = 0; d < sessions.size(); ++d) {
                    for (size_t k = 0; k < part[d].size(); ++k) work[origin[d][k]] = std::move(part[d][k]);
                    for (int q = 0; q < 4; ++q) tm[q] = std::max(tm[q], tms[d][q]);        // (the devices run side by side: the slowest one counts)
                }
            }
            t_s1 += tm[0]; t_stitch += tm[1]; t_s2 += tm[2]; t_stats += tm[3];
            // ---- writers, one contig per task; the contig table in contig order ----------------------------------------------------
            t0 = now_s();
            std::vector<std::string> rows(work.size());
            parallel_for(work.size(), n_threads, [&](size_t i) {
                ContigWork& w = work[i];
                w.snpless = get_frags_in_snpless_gaps(w.final_ranges, *w.snp_to_genome_pos, w.frags_without_snps, o.block_length, w.all_frags);
                rows[i] = write_contig_files(w, o);
            });
            for (const std::string& r : rows) append_contig_ploidy_row(o, r);
            t_write += now_s() - t0;
            // the Frags of a batch are millions of small objects: freeing them goes to a thread of its own, the next batch's ingest does not wait for it
            if (freer.joinable()) freer.join();
            freer = std::thread([w = std::make_shared<std::vector<ContigWork>>(std::move(work))]() mutable { w.reset(); });
        }
        }       // (next segment of the BAM)
        fprintf(stderr, "BAM: %zu records in %zu segments, %.3fs of inflate + decode, largest inflated buffer %zu MiB\n", n_records, n_segments, t_stream, stream.peak_buffer_bytes() >> 20);
        fprintf(stderr, "Realignment: %zu calls scored on the device in %.3fs (inside the ingest time)\n", n_realign_device, t_realign);
        fprintf(stderr, "Batches %zu; ingest %.3fs, phasing (upload + S1 + graph) %.3fs, LP + paths %.3fs, S2 %.3fs, COV/ERR/HAPQ %.3fs, writers %.3fs\n", n_batches, t_ingest, t_s1,
                t_stitch, t_s2, t_stats, t_write);
        if (reference_arith && n_batches_orders)
            fprintf(stderr, "Arithmetic: %zu of %zu batches carried the set orders of fragments that are not one alignment's (%zu merged from several alignments, %zu cut down by "
                            "--ignore-monomorphic): replayed on the host, every batch phased in the reference's running sums\n", n_batches_orders, n_batches, n_frags_merged, n_frags_cut);
        if (lp_report)
        fprintf(stderr, "LP: the optimum is not unique for %zu of %zu contigs (%zu of %zu edge flows differ in some other optimal solution); this run used the '%s' vertex, "
                        "rerun with --lp-tie %s to see what depends on it\n", lp_not_unique.load(), lp_contigs.load(), lp_movable.load(), lp_edges.load(),
                lp_tie == LpTie::First ? "first" : "last", lp_tie == LpTie::First ? "last" : "first");
        fprintf(stderr, "Total time taken is %.3fs\n", now_s() - t_all);
        // everything is written and closed: leave without tearing down the records, maps and sequences one by one (seconds for a large BAM)
        if (dump.is_open()) dump.close();
        sessions.clear();
        // every output stream of this run was scoped to the function that wrote it and is closed; stdio is flushed here.  What is left are the records,
        // maps and sequences of the inputs, whose node-by-node teardown takes seconds for a large run: the process leaves without it (the freer of the
        // last batch is abandoned with it, on the success path only).
        fflush(nullptr);
        if (freer.joinable()) freer.detach();
        std::_Exit(0);
    } catch (const Error& e) {
        fprintf(stderr, "floria-hip: error: %s\n", e.what());
        return 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "floria-hip: error: %s\n", e.what());
        return 1;
    }
    return 0;
}

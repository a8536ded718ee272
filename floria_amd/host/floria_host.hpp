// floria_host.hpp — C++ host mirror of the two reference call sites, on top of the C ABI (include/floria_hip.h).
//
// The reference is compiled code (Rust) and there is no Rust toolchain in this image, so the host side above the C ABI is
// C++ with the reference's names, argument meaning and error behaviour:
//   floria::get_range_with_lengths          utils_frags.rs:405-463
//   floria::generate_hap_graph              graph_processing.rs:325-372   (get_local_hap_blocks per block -> process_chunks ->
//                                                                          update_hap_graph, all on the device)
//   floria::process_reads_for_final_parts   part_block_manip.rs:174-274
//   floria::solve_lp_graph                  solve_flow.rs:195-290          (host; exact min-cost flow, see stitch.cpp)
//   floria::get_disjoint_paths_rewrite      graph_processing.rs:462-750    (host)
//   floria::get_frags_in_snpless_gaps       part_block_manip.rs:622-675    (host)
//   floria::write_outputs                   file_writer.rs:21-84,151-165,308-369,699-993   (.vartigs, .haplosets, vartig_info.txt,
//                                                                          reads_without_snps.tsv, contig_ploidy_info.tsv)
//   floria::get_vcf_profile, get_contigs_to_phase, get_frags_from_bamvcf_rewrite, get_fasta_seqs   file_reader.rs (ingest.cpp)
// Where the reference panics or exits (malformed VCF positions, :422-425) these functions throw floria::Error carrying the
// library's message.  There is no CPU fallback: constructing a Session without a usable MI355X throws.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "../../include/floria_hip.h"

namespace floria {

typedef size_t GnPosition;          // types_structs.rs:11
typedef uint32_t SnpPosition;       // types_structs.rs:12
typedef uint8_t Genotype;           // types_structs.rs:13

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// The hot-path fields of `Options` (types_structs.rs:20-51) with the CLI defaults (parse_cmd_line.rs)
struct Options {
    std::string bam_file, vcf_file, reference_fasta, out_dir = "floria_out_dir";     // -b -v -r -o
    double epsilon = 0.04;              // -e (auto-estimated from the BAM when absent, parse_cmd_line.rs:72-90)
    size_t max_number_solns = 10;       // -n
    size_t max_ploidy = 5;              // -p
    size_t block_length = 15000;        // -l (auto-estimated when absent)
    double snp_density = 0.0005;        // -d
    bool stopping_heuristic = true;     // !--no-stop-heuristic
    uint8_t ploidy_sensitivity = 2;     // -s
    size_t num_threads = 10;            // -t (accepted; the blocks run on the GPU)
    size_t snp_count_filter = 100;      // --snp-count-filter
    uint8_t mapq_cutoff = 15;           // -m
    int64_t supp_aln_dist_cutoff = 40000;   // --supp-aln-dist-cutoff
    bool dont_use_supp_aln = false;     // -X
    bool overwrite = false;             // --overwrite
    std::vector<std::string> list_to_phase;   // -G
    bool reassign_short = false;        // hidden flag; not supported (part_block_manip.rs:235-270)
    bool output_reads = false;          // --output-reads: fastq of every haploset's reads (file_writer.rs:370-560)
    bool gzip = false;                  // --gzip-reads
    bool trim_reads = false;            // --extra-trimming
    bool ignore_monomorphic = false;    // --ignore-monomorphic (utils_frags.rs:713-772)
    int device = 0;
};

// The per-read maps of the reference's Frag (FxHashMap<SnpPosition, _>, types_structs.rs:72-84) as a sorted vector with std::map's
// interface: a read has ~100 entries, inserted in ascending order by the CIGAR walk — one allocation instead of one tree node per entry
// (std::map made the ingest of 200 k long reads spend its time in malloc/free: 53 M nodes), iteration ascends like the BTreeMap-like uses need.
template <class K, class V> class FlatMap {
public:
    typedef std::pair<K, V> value_type;
    typedef typename std::vector<value_type>::iterator iterator;
    typedef typename std::vector<value_type>::const_iterator const_iterator;
    typedef typename std::vector<value_type>::const_reverse_iterator const_reverse_iterator;
    iterator begin() { return v_.begin(); }
    iterator end() { return v_.end(); }
    const_iterator begin() const { return v_.begin(); }
    const_iterator end() const { return v_.end(); }
    const_reverse_iterator rbegin() const { return v_.rbegin(); }
    const_reverse_iterator rend() const { return v_.rend(); }
    size_t size() const { return v_.size(); }
    bool empty() const { return v_.empty(); }
    void clear() { v_.clear(); }
    iterator lower_bound(const K& k) { return std::lower_bound(v_.begin(), v_.end(), k, [](const value_type& a, const K& b) { return a.first < b; }); }
    const_iterator lower_bound(const K& k) const { return std::lower_bound(v_.begin(), v_.end(), k, [](const value_type& a, const K& b) { return a.first < b; }); }
    iterator upper_bound(const K& k) { return std::upper_bound(v_.begin(), v_.end(), k, [](const K& a, const value_type& b) { return a < b.first; }); }
    const_iterator upper_bound(const K& k) const { return std::upper_bound(v_.begin(), v_.end(), k, [](const K& a, const value_type& b) { return a < b.first; }); }
    iterator find(const K& k) { iterator it = lower_bound(k); return it != v_.end() && it->first == k ? it : v_.end(); }
    const_iterator find(const K& k) const { const_iterator it = lower_bound(k); return it != v_.end() && it->first == k ? it : v_.end(); }
    size_t count(const K& k) const { return find(k) != v_.end() ? 1 : 0; }
    V& operator[](const K& k) {
        if (v_.empty() || v_.back().first < k) { v_.emplace_back(k, V()); return v_.back().second; }
        iterator it = lower_bound(k);
        if (it != v_.end() && it->first == k) return it->second;
        return v_.insert(it, value_type(k, V()))->second;
    }
    V& at(const K& k) { iterator it = find(k); if (it == v_.end()) throw std::out_of_range("FlatMap::at"); return it->second; }
    const V& at(const K& k) const { const_iterator it = find(k); if (it == v_.end()) throw std::out_of_range("FlatMap::at"); return it->second; }
    iterator erase(iterator it) { return v_.erase(it); }
    size_t erase(const K& k) { iterator it = find(k); if (it == v_.end()) return 0; v_.erase(it); return 1; }
private:
    std::vector<value_type> v_;
};

// types_structs.rs:68-85 (the fields the hot path reads); ordered (flat) maps: positions ascend
struct Frag {
    std::string id;
    size_t counter_id = 0;
    FlatMap<SnpPosition, Genotype> seq_dict;
    FlatMap<SnpPosition, uint8_t> qual_dict;
    SnpPosition first_position = UINT32_MAX, last_position = 0;
    // the fields ingest fills for the writers (types_structs.rs:80-84)
    bool is_paired = false;
    bool merged_positions = false;                                       // `positions` was extended by a mate's / supplementary piece's set (file_reader.rs:541, 639): its iteration order is not that of one CIGAR walk
    GnPosition first_pos_base = SIZE_MAX, last_pos_base = SIZE_MAX;      // reference span of the alignment (0-based start, end exclusive)
    size_t seq_len[2] = {0, 0};                                          // bases of seq_string[0], seq_string[1]
    std::string seq_string[2];                                           // only kept with --output-reads: DnaString::from_acgt_bytes of SEQ (anything but ACGT -> A)
    std::vector<uint8_t> qual_string[2];                                 // QUAL + 33
    FlatMap<SnpPosition, std::pair<uint8_t, GnPosition>> snp_pos_to_seq_pos;
    // What `positions` (types_structs.rs:79, the FxHashSet the reference's distance loop iterates, utils_frags.rs:35) went through, kept only where it is NOT the
    // set of one CIGAR walk: the SNP positions of every alignment combine_frags merged into this fragment, in merge order (positions.extend(..), file_reader.rs:541
    // and :639; segment 0 = the receiving alignment, possibly empty), and the positions --ignore-monomorphic removed afterwards (utils_frags.rs:745-755).
    // positions_order() replays that on an emulated set; a Batch hands the result to the library as floria_pileup_packed::set_order for the reference-arithmetic mode.
    std::vector<std::vector<SnpPosition>> position_segments;
    std::vector<SnpPosition> removed_positions;
    bool other_set_order() const { return !position_segments.empty() || !removed_positions.empty(); }
    std::vector<SnpPosition> positions_order() const;                    // the positions in the iteration order of the set
    void update(SnpPosition snp_pos, Genotype geno, uint8_t qual) {      // update_frag, types_structs.rs:286-324
        seq_dict[snp_pos] = geno; qual_dict[snp_pos] = qual;
        if (snp_pos < first_position) first_position = snp_pos;
        if (snp_pos > last_position) last_position = snp_pos;
    }
    bool operator<(const Frag& o) const {                                 // Frag::cmp, types_structs.rs:87-93
        if (first_position != o.first_position) return first_position < o.first_position;
        if (last_position != o.last_position) return last_position > o.last_position;
        return counter_id < o.counter_id;
    }
};

// types_structs.rs:155-166
struct HapNode {
    std::vector<const Frag*> frag_set;                                   // ascending counter_id
    std::vector<std::pair<size_t, double>> out_edges, in_edges, out_flows;
    size_t column = SIZE_MAX, row = SIZE_MAX, id = SIZE_MAX;
    double cov = 0.0;
    std::pair<SnpPosition, SnpPosition> snp_endpoints;
};

// One context + the contig currently resident in HBM.
class Session {
public:
    explicit Session(int device = 0);
    ~Session();
    Session(const Session&) = delete;
    Session& operator=(const Session&) = delete;
    floria_hip_ctx* ctx() const { return ctx_; }
    // flatten + upload `all_frags` (must be sorted by Frag::cmp with counter_id == index, floria.rs:289-293)
    void load_contig(const std::vector<Frag>& all_frags);
    floria_hip_contig* contig() const { return contig_; }
    const std::vector<Frag>* frags() const { return frags_; }
private:
    floria_hip_ctx* ctx_ = nullptr;
    floria_hip_contig* contig_ = nullptr;
    const std::vector<Frag>* frags_ = nullptr;
};

// Contigs -> devices of one node: longest-processing-time-first on an estimated cost (the rule of floria_amd/shard.py, which bench.py broadcasts over
// RCCL): items in descending cost (ties: ascending index) go to the least loaded device (ties: the lowest).  Contigs share nothing, so the node-level
// parallelism of the reference (one rayon pool over the blocks of a contig, graph_processing.rs:345-362) becomes contigs dealt to GPUs.
std::vector<uint32_t> lpt_assign(const std::vector<double>& costs, uint32_t world);

std::vector<std::pair<SnpPosition, SnpPosition>> get_range_with_lengths(const std::vector<GnPosition>& snp_to_genome_pos, size_t block_length,
                                                                          size_t overlap_len, double minimal_density);

// graph_processing.rs:325-372.  `floria_out_dir` is only used by the reference's --debug dumps and is ignored here.
std::vector<std::vector<HapNode>> generate_hap_graph(Session& s, const std::vector<Frag>& all_frags, const std::vector<GnPosition>& snp_to_genome_pos,
                                                     const std::string& floria_out_dir, const Options& options);

// part_block_manip.rs:174-274.  Reads are visited in ascending counter_id unless `visit_order` (counter_ids) is given
// (the reference follows its FxHashMap's iteration order).
std::pair<std::vector<std::vector<const Frag*>>, std::vector<std::pair<SnpPosition, SnpPosition>>> process_reads_for_final_parts(
    Session& s, const std::vector<std::vector<const Frag*>>& all_joined_path_parts, const std::vector<Frag>& short_frags,
    const std::vector<std::pair<SnpPosition, SnpPosition>>& snp_range_parts_vec, const Options& options, const std::vector<GnPosition>& snp_to_genome_pos,
    const std::vector<uint32_t>* visit_order = nullptr);

// solve_flow.rs: FlowUpVec = Vec<((column, row), (column, row), flow)>
struct FlowUpdate { std::pair<size_t, size_t> n1, n2; double flow; };
typedef std::vector<FlowUpdate> FlowUpVec;
// The LP's optimum is usually not unique (stitch.cpp).  LpTie picks which optimal vertex the flow solver returns (the two extremes of its
// tie-breaking); LpInfo reports the optimal value and how many edges carry a different flow in some other optimal solution.
enum class LpTie { First, Last };
struct LpInfo { int64_t cost = 0; size_t movable_edges = 0; };
FlowUpVec solve_lp_graph(const std::vector<std::vector<HapNode>>& hap_graph, LpTie tie = LpTie::First, LpInfo* info = nullptr);
std::pair<std::vector<std::vector<const Frag*>>, std::vector<std::pair<SnpPosition, SnpPosition>>> get_disjoint_paths_rewrite(
    std::vector<std::vector<HapNode>>& hap_graph, const FlowUpVec& flow_update_vec, const Options& options);
std::vector<const Frag*> get_frags_in_snpless_gaps(const std::vector<std::pair<SnpPosition, SnpPosition>>& path_parts, const std::vector<GnPosition>& snp_to_gn_pos,
                                                   const std::vector<Frag>& snpless_frags, GnPosition block_len, const std::vector<Frag>& final_frags);

// file_writer.rs:21-84.  `out_bam_part_dir` is the contig's output directory (its path is part of every HAP header), `options.out_dir`
// holds contig_ploidy_info.tsv (appended to; the header is written by write_run_files).  --output-reads (fastq) is not supported.
void write_outputs(Session& s, const std::vector<std::vector<const Frag*>>& part, const std::vector<std::pair<SnpPosition, SnpPosition>>& snp_range_parts_vec,
                   const std::string& out_bam_part_dir, const std::string& prefix, const std::string& contig, const std::vector<GnPosition>& snp_pos_to_genome_pos,
                   const Options& options, const std::vector<const Frag*>& snpless_frags, size_t contig_len);
// parse_cmd_line.rs:116-135: create the output directory (it must not exist unless --overwrite), cmd.log, contig_ploidy_info.tsv header
void prepare_contig_dir(const std::string& contig_out_dir, const Options& options);       // --overwrite: remove_dir_all of an existing contig directory (floria.rs:271-281)
void write_run_files(const Options& options, int argc, char** argv, const std::string& note = std::string());    // note: a second line of cmd.log

// ---- ingest (file_reader.rs) ---------------------------------------------------------------------------------------------------
struct VcfProfile {          // types_structs.rs:53-58, per contig; plus get_genotypes_from_vcf_hts' position vectors (:113-175)
    std::map<std::string, std::map<GnPosition, std::vector<Genotype>>> vcf_pos_allele_map;
    std::map<std::string, std::map<GnPosition, SnpPosition>> vcf_pos_to_snp_counter_map;
    std::map<std::string, std::map<SnpPosition, GnPosition>> vcf_snp_pos_to_gn_pos_map;
    std::map<std::string, std::vector<GnPosition>> snp_to_genome_pos;
};
// A record is a set of views into the inflated BAM stream the BamFile keeps alive: nothing is copied or unpacked per record
// (200 k long reads = 2 GB of bases; a per-record std::string each made the parallel decode slower than the serial one).
struct BamSeqView {                    // SEQ, 4 bits per base
    const unsigned char* packed = nullptr; size_t n = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    char operator[](size_t k) const { const unsigned char b = packed[k >> 1]; return "=ACMGRSVTWYHKDBN"[(k & 1) ? (b & 15) : (b >> 4)]; }
};
struct BamBytesView {                  // QUAL
    const uint8_t* p = nullptr; size_t n = 0;
    size_t size() const { return n; }
    uint8_t operator[](size_t k) const { return p[k]; }
};
struct BamCigarView {                  // n x u32 (len << 4 | op), possibly unaligned
    const unsigned char* p = nullptr; size_t n = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    uint32_t operator[](size_t k) const { uint32_t v; memcpy(&v, p + 4 * k, 4); return v; }
};
struct BamRecord {
    int32_t tid = -1, pos = 0;
    uint8_t mapq = 0;
    uint16_t flags = 0;
    std::string_view qname;
    BamCigarView cigar;
    BamSeqView seq;                    // ASCII bases through operator[]
    BamBytesView qual;
};
struct BamFile {
    std::vector<unsigned char> raw;    // the inflated stream (the records point into it: a BamFile may be moved, not copied)
    std::vector<std::string> target_names;
    std::vector<uint64_t> target_len;
    std::vector<BamRecord> records;    // file order
    std::vector<std::vector<uint32_t>> by_tid;     // record indices per target, file order (what an indexed fetch() of the contig yields)
    int32_t tid_begin = 0, tid_end = 0;            // BamStream segments: the targets [tid_begin, tid_end) are COMPLETE in this segment (read_bam: all of them)
    BamFile() = default;
    BamFile(BamFile&&) = default;
    BamFile& operator=(BamFile&&) = default;
    BamFile(const BamFile&) = delete;
    BamFile& operator=(const BamFile&) = delete;
};
// A coordinate-sorted BAM in bounded memory: the file is mapped, its BGZF members are indexed from their headers (no inflation), and next() inflates
// (in parallel) just enough members to return the records of the next run of COMPLETE targets — at least `min_bytes` of inflated records, more only when one
// target alone is larger.  The reference reaches the same records through the .bai index, one contig at a time (file_reader.rs:389-436); no index is
// needed here because targets come in header order in a sorted file.  A file that is not sorted by target is refused, a gzip file without BGZF members
// is read whole (one segment).
class BamStream {
public:
    BamStream(const std::string& path, size_t threads = 1);
    ~BamStream();
    BamStream(const BamStream&) = delete;
    BamStream& operator=(const BamStream&) = delete;
    const std::vector<std::string>& target_names() const;
    bool next(BamFile& segment, size_t min_bytes);     // false: end of file (every target has been handed out)
    void rewind();
    size_t peak_buffer_bytes() const;                  // largest inflated buffer so far (what "bounded" means for this file)
private:
    struct Impl;
    std::unique_ptr<Impl> p_;
};
BamFile read_bam(const std::string& path, size_t threads = 1);                         // BGZF + BAM, whole file (no index needed); members and records decode in parallel
std::vector<std::string> get_contigs_to_phase(const BamFile& bam);                     // file_reader.rs:738-746
VcfProfile get_vcf_profile(const std::string& vcf_file, const std::vector<std::string>& ref_chroms);       // :239-314 (+ :113-175); text VCF, optionally gzipped
std::map<std::string, std::string> get_fasta_seqs(const std::string& fasta_file);      // :462-489 (whole sequences)
// :343-460 + combine_frags :491-659 + frag_from_record :661-736; with `ref_seq` (the contig's reference sequence) every call is
// realigned (alignment.rs:7-64, exact affine-gap DP in place of block-aligner)
std::pair<std::vector<Frag>, std::vector<Frag>> get_frags_from_bamvcf_rewrite(const BamFile& bam, const VcfProfile& vcf_profile, const Options& options, const std::string& contig,
                                                                               const std::string* ref_seq = nullptr);
// The same in two halves with the realignment's DP on the device between them.  alignment::realign decides most calls from the mismatch
// count of the two 32-base windows (an exact shortcut); the windows it cannot decide are queued — read window, reference window,
// candidate alleles, and the place the winning allele goes — scored by floria_hip_realign for a whole batch of contigs at once
// (Session::realign), and only then are mates and supplementary pieces merged (combine_frags copies the calls).
struct RealignQueue {
    std::vector<uint8_t> read_windows, ref_windows, alleles, n_alleles;    // 32 / 32 / FLORIA_MAX_ALLELES / 1 bytes per call
    std::vector<Genotype*> dst;
    size_t size() const { return dst.size(); }
    void append(RealignQueue&& other);
};
class ContigIngest {
public:
    // records of `contig` -> one Frag per passing alignment (file_reader.rs:343-430); with a queue the undecided realignments are deferred
    ContigIngest(const BamFile& bam, const VcfProfile& vcf_profile, const Options& options, const std::string& contig, const std::string* ref_seq, RealignQueue* queue);
    ~ContigIngest();
    ContigIngest(ContigIngest&&) noexcept;
    ContigIngest& operator=(ContigIngest&&) noexcept;
    std::pair<std::vector<Frag>, std::vector<Frag>> finish();              // combine_frags (:491-659) -> (frags with SNPs, frags without)
private:
    struct Impl;
    std::unique_ptr<Impl> p_;
};
std::pair<size_t, double> l_epsilon_auto_detect(const BamFile& bam);                   // :749-826
// The same fed segment by segment (complete targets in header order, as BamStream::next yields them): done() once 1000 columns have been sampled,
// which is where the reference's pileup loop breaks as well.
class EpsilonEstimator {
public:
    EpsilonEstimator();
    ~EpsilonEstimator();
    void feed(const BamFile& segment);
    bool done() const;
    std::pair<size_t, double> result();                // (block length, epsilon)
private:
    struct Impl;
    std::unique_ptr<Impl> p_;
};

// part_block_manip.rs:517-616: (hapqs, rel_err per haploset, avg_err) — the HAPQ / REL_ERR header fields and the contig table's avg_err.
struct HapqResult { std::vector<uint8_t> hapqs; std::vector<double> rel_err; double avg_err = 0.0; };

// ---- many contigs at once ---------------------------------------------------------------------------------------------------------------
// The reference walks its contigs serially (floria.rs:229) because its parallelism is inside a contig (rayon over blocks).  On the
// GPU the throughput comes from thousands of blocks in flight, and a metagenome has thousands of small contigs: a host takes a batch
// of contigs through every stage together — one pipelined upload + S1 call, one hap-graph call, the LP / path peeling per contig
// on host threads, one S2 call, one COV/ERR and one HAPQ call — and writes the files contig by contig.  Results per contig are
// exactly those of the per-contig functions above.
struct ContigWork {
    std::string name, out_dir;
    std::vector<Frag> all_frags, frags_without_snps;                   // all_frags sorted, counter_id == index (floria.rs:289-293)
    const std::vector<GnPosition>* snp_to_genome_pos = nullptr;
    size_t contig_len = 0;
    // filled stage by stage
    std::vector<std::pair<SnpPosition, SnpPosition>> iter_vec;         // get_range_with_lengths
    std::vector<std::vector<HapNode>> hap_graph;                       // generate_hap_graph
    FlowUpVec flows;                                                   // solve_lp_graph
    std::vector<std::vector<const Frag*>> path_parts, final_parts;     // get_disjoint_paths_rewrite / process_reads_for_final_parts
    std::vector<std::pair<SnpPosition, SnpPosition>> path_ranges, final_ranges;
    std::vector<const Frag*> snpless;                                  // get_frags_in_snpless_gaps
    std::vector<double> stats;                                         // 4 per final haploset: cov, err, total_err, total_cov
    HapqResult hq;
};
class Batch {
public:
    // marshals the Frags into the pinned compact wire form (floria_pileup_packed); with_set_orders: contigs that hold merged or cut-down fragments also get
    // the iteration orders of their fragments' position sets (set_order: what the reference-arithmetic mode needs to add in the reference's order)
    Batch(Session& s, std::vector<ContigWork>& work, bool with_set_orders = false);
    ~Batch();
    Batch(const Batch&) = delete;
    Batch& operator=(const Batch&) = delete;
    void generate_hap_graphs(const Options& options);                  // upload + S1 + hap graph for every contig (graph_processing.rs:325-372)
    void process_reads_for_final_parts(const Options& options);        // S2 for every contig, from path_parts / path_ranges
    void stats_and_hapq(const Options& options);                       // get_errors_cov_from_frags + get_hapq for every final haploset
private:
    Session& s_;
    std::vector<ContigWork>& work_;
    void* pinned_ = nullptr;
    std::vector<floria_pileup_packed> piles_;
    std::vector<floria_hip_contig*> handles_;
};
// utils_frags::remove_monomorphic_allele (utils_frags.rs:713-772, --ignore-monomorphic): SNPs where one allele carries (almost) all of the
// phred weight are dropped from every read; reads left without SNPs are dropped; the rest is sorted and renumbered
std::vector<Frag> remove_monomorphic_allele(std::vector<Frag> frags, double error);
// write_reads + write_nosnp_reads (file_writer.rs:86-150, 370-560, --output-reads): long_reads/{i}_part.fastq, short_reads/{i}_part_paired{1,2}.fastq, snpless*.fastq
void write_reads(const std::vector<std::vector<const Frag*>>& part, const std::vector<std::pair<SnpPosition, SnpPosition>>& snp_range_parts_vec, const std::string& out_bam_part_dir,
                 bool extend_read_clipping, const std::vector<uint8_t>& hapqs, bool gzip);
void write_nosnp_reads(const std::string& out_bam_part_dir, const std::vector<const Frag*>& snpless_frags, bool gzip);
// scores the queued realignment windows on the device (floria_hip_realign) and stores the winning alleles where they belong
void realign_queue_on_device(Session& s, RealignQueue& queue);
// write_outputs for a contig whose statistics were computed by Batch::stats_and_hapq
void write_outputs(const ContigWork& w, const Options& options);
// the same in two halves, for hosts that write the contigs of a batch from several threads: the files of the contig (returns its
// contig_ploidy_info.tsv row), and the append of the rows in contig order
std::string write_contig_files(const ContigWork& w, const Options& options);
void append_contig_ploidy_row(const Options& options, const std::string& row);
HapqResult get_hapq(Session& s, const std::vector<std::vector<const Frag*>>& parts, const std::vector<GnPosition>& snp_to_genome_pos,
                    const std::vector<std::pair<SnpPosition, SnpPosition>>& snp_range_parts_vec, const Options& options);

}  // namespace floria

// floria_host.hpp — C++ host mirror of the two reference call sites, on top of the C ABI (include/floria_hip.h).
//
// The reference is compiled code (Rust) and there is no Rust toolchain in this image, so the host side above the C ABI is
// C++ with the reference's names, argument meaning and error behaviour:
//   floria::get_range_with_lengths          utils_frags.rs:405-463
//   floria::generate_hap_graph              graph_processing.rs:325-372   (get_local_hap_blocks per block -> process_chunks ->
//                                                                          update_hap_graph, all on the device)
//   floria::process_reads_for_final_parts   part_block_manip.rs:174-274
// Where the reference panics or exits (malformed VCF positions, :422-425) these functions throw floria::Error carrying the
// library's message.  There is no CPU fallback: constructing a Session without a usable MI355X throws.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
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
    double epsilon = 0.04;
    size_t max_number_solns = 10;       // -n
    size_t max_ploidy = 5;              // -p
    size_t block_length = 15000;        // -l
    double snp_density = 0.0005;        // -d
    bool stopping_heuristic = true;
    uint8_t ploidy_sensitivity = 2;     // -s
    bool reassign_short = false;        // hidden flag; not supported (part_block_manip.rs:235-270)
    int device = 0;
};

// types_structs.rs:68-85 (the fields the hot path reads); ordered maps: positions ascend
struct Frag {
    std::string id;
    size_t counter_id = 0;
    std::map<SnpPosition, Genotype> seq_dict;
    std::map<SnpPosition, uint8_t> qual_dict;
    SnpPosition first_position = UINT32_MAX, last_position = 0;
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
    std::vector<std::pair<size_t, double>> out_edges, in_edges;
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

// part_block_manip.rs:517-616: (hapqs, rel_err per haploset, avg_err) — the HAPQ / REL_ERR header fields and the contig table's avg_err.
struct HapqResult { std::vector<uint8_t> hapqs; std::vector<double> rel_err; double avg_err; };
HapqResult get_hapq(Session& s, const std::vector<std::vector<const Frag*>>& parts, const std::vector<GnPosition>& snp_to_genome_pos,
                    const std::vector<std::pair<SnpPosition, SnpPosition>>& snp_range_parts_vec, const Options& options);

}  // namespace floria

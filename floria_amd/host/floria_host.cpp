// floria_host.cpp — see floria_host.hpp.  Pure C++17 client of libfloria_hip.so's C ABI (no HIP headers needed here).
#include "floria_host.hpp"

#include <algorithm>

namespace floria {

namespace {
void check(int rc) { if (rc != 0) throw Error(rc, floria_hip_last_error()); }
constexpr double MIN_SHARED_READS_UNAMBIG = 2.;     // constants.rs:4
}  // namespace

Session::Session(int device) { check(floria_hip_create(device, &ctx_)); }
Session::~Session() {
    if (contig_) floria_hip_contig_free(contig_);
    if (ctx_) floria_hip_destroy(ctx_);
}

void Session::load_contig(const std::vector<Frag>& all_frags) {
    if (contig_) { floria_hip_contig_free(contig_); contig_ = nullptr; }
    std::vector<uint32_t> off{0}, snp, first, last;
    std::vector<uint8_t> al, q;
    for (size_t i = 0; i < all_frags.size(); ++i) {
        const Frag& f = all_frags[i];
        if (f.counter_id != i) throw Error(FLORIA_E_INVALID, "all_frags must be sorted with counter_id == index (floria.rs:289-293)");
        for (const auto& kv : f.seq_dict) { snp.push_back(kv.first); al.push_back(kv.second); q.push_back(f.qual_dict.at(kv.first)); }
        off.push_back((uint32_t)snp.size()); first.push_back(f.first_position); last.push_back(f.last_position);
    }
    floria_pileup p{off.data(), snp.data(), al.data(), q.data(), first.data(), last.data(), (uint32_t)all_frags.size()};
    check(floria_hip_contig_upload(ctx_, &p, &contig_));
    frags_ = &all_frags;
}

std::vector<std::pair<SnpPosition, SnpPosition>> get_range_with_lengths(const std::vector<GnPosition>& g, size_t block_length, size_t overlap_len,
                                                                          double minimal_density) {
    std::vector<uint64_t> g64(g.begin(), g.end());
    floria_ranges* r = nullptr;
    check(floria_hip_block_ranges(g64.data(), (uint32_t)g64.size(), block_length, overlap_len, minimal_density, &r));
    std::vector<std::pair<SnpPosition, SnpPosition>> out(r->n);
    for (uint32_t i = 0; i < r->n; ++i) out[i] = {r->start[i], r->end[i]};
    floria_hip_ranges_free(r);
    return out;
}

std::vector<std::vector<HapNode>> generate_hap_graph(Session& s, const std::vector<Frag>& all_frags, const std::vector<GnPosition>& snp_to_genome_pos,
                                                     const std::string&, const Options& o) {
    if (s.frags() != &all_frags) s.load_contig(all_frags);
    const auto iter_vec = get_range_with_lengths(snp_to_genome_pos, o.block_length, o.block_length / 3, o.snp_density);     // :334-339
    std::vector<uint32_t> bs, be;
    for (auto& r : iter_vec) { bs.push_back(r.first); be.push_back(r.second); }
    floria_params prm{o.epsilon, (uint32_t)o.max_ploidy, (uint32_t)o.max_number_solns, o.ploidy_sensitivity, o.stopping_heuristic ? 1 : 0};
    floria_block_result* res = nullptr;
    check(floria_hip_phase_blocks_resident(s.ctx(), s.contig(), bs.data(), be.data(), (uint32_t)bs.size(), &prm, &res));     // :345-362
    floria_hap_graph* hg = nullptr;
    int rc = floria_hip_hap_graph(s.ctx(), res, &hg);                                                                       // :369 update_hap_graph
    if (rc) { floria_hip_block_result_free(res); check(rc); }
    // process_chunks (:306-323): columns = blocks that returned Some, in block order; ids run over all nodes
    std::vector<std::vector<HapNode>> cols;
    std::vector<size_t> col_of_block(res->n_blocks, SIZE_MAX);
    size_t id_counter = 0;
    for (uint32_t b = 0; b < res->n_blocks; ++b) {
        const uint32_t p = res->best_ploidy[b];
        if (p == 0) continue;
        std::vector<HapNode> col(p);
        for (uint64_t i = res->read_off[b]; i < res->read_off[b + 1]; ++i) col[res->part[i]].frag_set.push_back(&all_frags[res->read_id[i]]);
        for (uint32_t k = 0; k < p; ++k) {
            col[k].row = k; col[k].column = cols.size(); col[k].id = id_counter++;
            col[k].snp_endpoints = iter_vec[b];
            col[k].cov = hg->node_cov[hg->node_off[b] + k];
        }
        col_of_block[b] = cols.size();
        cols.push_back(std::move(col));
    }
    // update_hap_graph (:22-100): edges with weight >= MIN_SHARED_READS_UNAMBIG (:51), mirrored into in_edges (:87-96)
    for (uint32_t b = 0; b < res->n_blocks; ++b) {
        const int32_t pb = hg->pred[b];
        if (pb < 0 || res->best_ploidy[b] == 0) continue;
        auto& c1 = cols[col_of_block[pb]]; auto& c2 = cols[col_of_block[b]];
        const uint32_t p1 = res->best_ploidy[pb], p2 = res->best_ploidy[b];
        for (uint32_t j = 0; j < p1; ++j)
            for (uint32_t l = 0; l < p2; ++l) {
                const double w = (double)hg->edge_w[hg->edge_off[b] + (uint64_t)j * p2 + l];
                if (w >= MIN_SHARED_READS_UNAMBIG) { c1[j].out_edges.push_back({l, w}); c2[l].in_edges.push_back({j, w}); }
            }
    }
    floria_hip_hap_graph_free(hg);
    floria_hip_block_result_free(res);
    return cols;
}

std::pair<std::vector<std::vector<const Frag*>>, std::vector<std::pair<SnpPosition, SnpPosition>>> process_reads_for_final_parts(
    Session& s, const std::vector<std::vector<const Frag*>>& parts, const std::vector<Frag>&, const std::vector<std::pair<SnpPosition, SnpPosition>>& ranges,
    const Options& o, const std::vector<GnPosition>&, const std::vector<uint32_t>* visit_order) {
    if (o.reassign_short) throw Error(FLORIA_E_UNSUPPORTED, "--reassign-short (hidden flag) is not supported");
    if (!s.contig() || !s.frags()) throw Error(FLORIA_E_INVALID, "no contig loaded");
    std::vector<uint64_t> off{0};
    std::vector<uint32_t> reads, rng;
    for (size_t g = 0; g < parts.size(); ++g) {
        for (const Frag* f : parts[g]) reads.push_back((uint32_t)f->counter_id);
        off.push_back(reads.size());
        rng.push_back(ranges[g].first); rng.push_back(ranges[g].second);
    }
    floria_groups* out = nullptr;
    check(floria_hip_reassign_ordered(s.ctx(), s.contig(), off.data(), reads.data(), rng.data(), (uint32_t)parts.size(),
                                      visit_order ? visit_order->data() : nullptr, visit_order ? (uint32_t)visit_order->size() : 0, o.epsilon, &out));
    std::vector<std::vector<const Frag*>> np(out->n_groups);
    std::vector<std::pair<SnpPosition, SnpPosition>> nr(out->n_groups);
    for (uint32_t g = 0; g < out->n_groups; ++g) {
        for (uint64_t i = out->grp_off[g]; i < out->grp_off[g + 1]; ++i) np[g].push_back(&(*s.frags())[out->grp_read[i]]);
        nr[g] = {out->range[2 * g], out->range[2 * g + 1]};
    }
    floria_hip_groups_free(out);
    return {std::move(np), std::move(nr)};
}

HapqResult get_hapq(Session& s, const std::vector<std::vector<const Frag*>>& parts, const std::vector<GnPosition>& snp_to_genome_pos,
                    const std::vector<std::pair<SnpPosition, SnpPosition>>& ranges, const Options& o) {
    if (!s.contig() || !s.frags()) throw Error(FLORIA_E_INVALID, "no contig loaded");
    std::vector<uint64_t> off{0};
    std::vector<uint32_t> reads, rng;
    for (size_t g = 0; g < parts.size(); ++g) {
        for (const Frag* f : parts[g]) reads.push_back((uint32_t)f->counter_id);
        off.push_back(reads.size());
        rng.push_back(ranges[g].first); rng.push_back(ranges[g].second);
    }
    std::vector<uint64_t> pos(snp_to_genome_pos.begin(), snp_to_genome_pos.end());
    HapqResult r;
    r.hapqs.assign(parts.size(), 0); r.rel_err.assign(parts.size(), 0.0); r.avg_err = 0.0;
    check(floria_hip_hapq(s.ctx(), s.contig(), off.data(), reads.data(), rng.data(), (uint32_t)parts.size(), pos.data(), (uint32_t)pos.size(),
                          (uint64_t)o.block_length, r.hapqs.data(), r.rel_err.data(), &r.avg_err));
    return r;
}

}  // namespace floria

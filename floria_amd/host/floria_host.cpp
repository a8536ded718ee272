// floria_host.cpp — see floria_host.hpp.  Pure C++17 client of libfloria_hip.so's C ABI (no HIP headers needed here).
#include "floria_host.hpp"

#include <algorithm>
#include <cmath>
#include <set>

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
    // fragments whose position set is not that of one CIGAR walk (merged mates / supplementary pieces, --ignore-monomorphic): the contig carries the replayed orders
    std::vector<uint32_t> set_order;
    bool other = false;
    for (const Frag& f : all_frags) other = other || f.other_set_order();
    if (other)
        for (const Frag& f : all_frags)
            for (SnpPosition sp : f.positions_order()) {
                const auto it = f.seq_dict.find(sp);
                if (it == f.seq_dict.end()) throw Error(FLORIA_E_INVALID, "Frag whose position provenance does not match its calls");
                set_order.push_back((uint32_t)(it - f.seq_dict.begin()));
            }
    if (other && set_order.size() != snp.size()) throw Error(FLORIA_E_INVALID, "Frag whose position provenance does not match its calls");
    floria_pileup p{off.data(), snp.data(), al.data(), q.data(), first.data(), last.data(), (uint32_t)all_frags.size(), other ? set_order.data() : nullptr};
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

namespace {
// process_chunks (graph_processing.rs:306-323) + update_hap_graph's edge lists (:22-100) for the blocks [b0, b1) of one contig out of a batch
// result: columns = blocks that returned Some, in block order; ids run over the contig's nodes; edges with weight >= MIN_SHARED_READS_UNAMBIG (:51)
std::vector<std::vector<HapNode>> build_columns(const floria_block_result* res, const floria_hap_graph* hg, uint32_t b0, uint32_t b1, const std::vector<Frag>& all_frags,
                                                const std::vector<std::pair<SnpPosition, SnpPosition>>& iter_vec) {
    std::vector<std::vector<HapNode>> cols;
    std::vector<size_t> col_of_block(b1 - b0, SIZE_MAX);
    size_t id_counter = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        const uint32_t p = res->best_ploidy[b];
        if (p == 0) continue;
        std::vector<HapNode> col(p);
        for (uint64_t i = res->read_off[b]; i < res->read_off[b + 1]; ++i) col[res->part[i]].frag_set.push_back(&all_frags[res->read_id[i]]);
        for (uint32_t k = 0; k < p; ++k) {
            col[k].row = k; col[k].column = cols.size(); col[k].id = id_counter++;
            col[k].snp_endpoints = iter_vec[b - b0];
            col[k].cov = hg->node_cov[hg->node_off[b] + k];
        }
        col_of_block[b - b0] = cols.size();
        cols.push_back(std::move(col));
    }
    for (uint32_t b = b0; b < b1; ++b) {
        const int32_t pb = hg->pred[b];
        if (pb < 0 || res->best_ploidy[b] == 0) continue;
        auto& c1 = cols[col_of_block[(uint32_t)pb - b0]]; auto& c2 = cols[col_of_block[b - b0]];
        const uint32_t p1 = res->best_ploidy[pb], p2 = res->best_ploidy[b];
        for (uint32_t j = 0; j < p1; ++j)
            for (uint32_t l = 0; l < p2; ++l) {
                const double w = (double)hg->edge_w[hg->edge_off[b] + (uint64_t)j * p2 + l];
                if (w >= MIN_SHARED_READS_UNAMBIG) { c1[j].out_edges.push_back({l, w}); c2[l].in_edges.push_back({j, w}); }
            }
    }
    return cols;
}
}  // namespace

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
    auto cols = build_columns(res, hg, 0, res->n_blocks, all_frags, iter_vec);
    floria_hip_hap_graph_free(hg);
    floria_hip_block_result_free(res);
    return cols;
}

namespace {
// std's FxHashSet<SnpPosition> as far as Frag.positions uses it: hashbrown's open-addressing table (x86-64 builds: probe groups of 16 control bytes at triangular
// strides, 7/8 load factor, growth by re-insertion in bucket order) under fxhash 0.2.1 (key * 0x517cc1b727220a95: low bits pick the first group, the top 7 bits
// tag the control byte).  Written for this one purpose — reserve, insert (which reserves room for one key BEFORE it looks the key up), removal, iteration in bucket
// order — and checked against the oracle's table and an independent bucket-list model by tests/test_host_cpu.py / tests/test_order_emulation.py.
class PositionSet {
public:
    size_t size() const { return n_; }
    void reserve(size_t additional) { if (additional > room_) regrow(std::max(n_ + additional, usable(nb_) + 1)); }
    void insert(SnpPosition k) {
        reserve(1);
        if (lookup(k) != SIZE_MAX) return;
        const size_t at = first_free(k);
        tag_[at] = (uint8_t)(hash(k) >> 57); key_[at] = k; ++n_; --room_;
        mirror(at);
    }
    void remove(SnpPosition k) {                         // (nothing is inserted after a removal on this path: whether the bucket becomes EMPTY or a tombstone changes nothing that is observed)
        const size_t at = lookup(k);
        if (at == SIZE_MAX) return;
        tag_[at] = GONE; --n_; mirror(at);
    }
    template <class F> void each(F f) const { for (size_t i = 0; i < nb_; ++i) if (!(tag_[i] & 0x80)) f(key_[i]); }
    void extend(const PositionSet& o) {                  // hashbrown's Extend: the whole size hint into an empty table, half of it (rounded up) otherwise
        reserve(n_ == 0 ? o.n_ : (o.n_ + 1) / 2);
        o.each([&](SnpPosition k) { insert(k); });
    }
private:
    static constexpr uint8_t FREE = 0xFF, GONE = 0x80;
    static constexpr size_t G = 16;
    std::vector<uint8_t> tag_;                           // nb_ + G control bytes: the first G repeated behind the last bucket
    std::vector<SnpPosition> key_;
    size_t nb_ = 0, n_ = 0, room_ = 0;
    static uint64_t hash(SnpPosition k) { return (uint64_t)k * 0x517cc1b727220a95ull; }
    static size_t usable(size_t nb) { return nb == 0 ? 0 : (nb <= 8 ? nb - 1 : nb / 8 * 7); }
    static size_t buckets_for(size_t cap) { if (cap < 4) return 4; if (cap < 8) return 8; size_t b = 1; while (b < cap * 8 / 7) b <<= 1; return b; }
    void mirror(size_t at) { tag_[((at - G) & (nb_ - 1)) + G] = tag_[at]; }
    size_t first_free(SnpPosition k) const {
        const size_t mask = nb_ - 1;
        for (size_t at = (size_t)hash(k) & mask, step = 0;; step += G, at = (at + step) & mask)
            for (size_t j = 0; j < G; ++j)
                if (tag_[at + j] & 0x80) {
                    size_t hit = (at + j) & mask;
                    if (!(tag_[hit] & 0x80)) { hit = 0; while (!(tag_[hit] & 0x80)) ++hit; }      // a table smaller than a group: the free byte was padding, the real one is in group 0
                    return hit;
                }
    }
    size_t lookup(SnpPosition k) const {
        if (nb_ == 0) return SIZE_MAX;
        const size_t mask = nb_ - 1;
        const uint8_t t = (uint8_t)(hash(k) >> 57);
        for (size_t at = (size_t)hash(k) & mask, step = 0;; step += G, at = (at + step) & mask) {
            bool open = false;
            for (size_t j = 0; j < G; ++j) {
                if (tag_[at + j] == t && key_[(at + j) & mask] == k) return (at + j) & mask;
                open = open || tag_[at + j] == FREE;
            }
            if (open) return SIZE_MAX;
        }
    }
    void regrow(size_t capacity) {
        PositionSet g;
        g.nb_ = buckets_for(capacity); g.tag_.assign(g.nb_ + G, FREE); g.key_.assign(g.nb_, 0); g.room_ = usable(g.nb_);
        for (size_t i = 0; i < nb_; ++i) if (!(tag_[i] & 0x80)) {
            const size_t at = g.first_free(key_[i]);
            g.tag_[at] = (uint8_t)(hash(key_[i]) >> 57); g.key_[at] = key_[i]; g.mirror(at); ++g.n_; --g.room_;
        }
        *this = std::move(g);
    }
};
// frag_from_record (file_reader.rs:661-733): seq_dict grows as the CIGAR walk inserts ascending SNP positions; positions = seq_dict.keys().collect()
PositionSet collected_positions(const std::vector<SnpPosition>& ascending) {
    PositionSet seq_dict, positions;
    for (SnpPosition k : ascending) seq_dict.insert(k);
    positions.extend(seq_dict);                         // (HashSet::from_iter is `extend` on an empty set)
    return positions;
}
}  // namespace

std::vector<SnpPosition> Frag::positions_order() const {
    PositionSet acc;
    if (position_segments.empty()) {
        std::vector<SnpPosition> keys;
        for (const auto& kv : seq_dict) keys.push_back(kv.first);
        acc = collected_positions(keys);
    } else {
        for (size_t i = 0; i < position_segments.size(); ++i) {
            PositionSet s = collected_positions(position_segments[i]);
            if (i == 0) acc = std::move(s); else acc.extend(s);
        }
    }
    for (SnpPosition k : removed_positions) acc.remove(k);
    std::vector<SnpPosition> out;
    acc.each([&](SnpPosition k) { out.push_back(k); });
    return out;
}

// utils_frags::remove_monomorphic_allele (utils_frags.rs:713-772).  phred_scale = 1f32 - 10f32^(-q/10) widened to f64 (:702-711); the
// per-(SNP, allele) sums only add such weights, all multiples of 2^-24: exact in any order, so the hash-map order of the reference does not matter.
std::vector<Frag> remove_monomorphic_allele(std::vector<Frag> frags, double error) {
    std::map<SnpPosition, std::map<Genotype, double>> allele_count_map;
    for (const Frag& f : frags)
        for (const auto& kv : f.seq_dict) {
            const float prob = 1.0f - powf(10.0f, (float)f.qual_dict.at(kv.first) / -10.0f);
            allele_count_map[kv.first][kv.second] += (double)prob;
        }
    std::set<SnpPosition> mono;
    for (const auto& am : allele_count_map) {
        if (am.second.size() == 1) { mono.insert(am.first); continue; }
        std::vector<double> vals;
        for (const auto& kv : am.second) vals.push_back(kv.second);
        std::sort(vals.begin(), vals.end(), [](double a, double b) { return a > b; });
        if (vals[0] * error > vals[1]) mono.insert(am.first);
    }
    std::vector<Frag> out;
    for (Frag& f : frags) {
        bool cut = false;
        for (const auto& kv : f.seq_dict) cut = cut || mono.count(kv.first) != 0;
        if (cut && f.position_segments.empty()) {            // `positions` keeps the layout it had with the removed keys in it (:745-755): remember them
            f.position_segments.emplace_back();
            for (const auto& kv : f.seq_dict) f.position_segments[0].push_back(kv.first);
        }
        for (auto it = f.seq_dict.begin(); it != f.seq_dict.end();) {
            if (mono.count(it->first)) { f.removed_positions.push_back(it->first); f.qual_dict.erase(it->first); f.snp_pos_to_seq_pos.erase(it->first); it = f.seq_dict.erase(it); } else ++it;
        }
        if (f.seq_dict.empty()) continue;
        f.first_position = f.seq_dict.begin()->first; f.last_position = f.seq_dict.rbegin()->first;
        out.push_back(std::move(f));
    }
    std::sort(out.begin(), out.end());
    for (size_t i = 0; i < out.size(); ++i) out[i].counter_id = i;
    return out;
}

void realign_queue_on_device(Session& s, RealignQueue& q) {
    const size_t n = q.size();
    if (!n) return;
    std::vector<uint8_t> best(n);
    check(floria_hip_realign(s.ctx(), q.read_windows.data(), q.ref_windows.data(), q.alleles.data(), q.n_alleles.data(), (uint64_t)n, best.data(), nullptr));
    for (size_t i = 0; i < n; ++i) *q.dst[i] = (Genotype)best[i];
    q = RealignQueue();
}

std::vector<uint32_t> lpt_assign(const std::vector<double>& costs, uint32_t world) {
    std::vector<size_t> order(costs.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return costs[a] > costs[b]; });
    std::vector<double> load(std::max<uint32_t>(1, world), 0.0);
    std::vector<uint32_t> owner(costs.size(), 0);
    for (size_t i : order) {
        uint32_t r = 0;
        for (uint32_t d = 1; d < load.size(); ++d) if (load[d] < load[r]) r = d;      // first minimum
        owner[i] = r; load[r] += costs[i];
    }
    return owner;
}

// ---- Batch: every device stage once for MANY contigs ----------------------------------------------------------------------------------
Batch::Batch(Session& s, std::vector<ContigWork>& work, bool with_set_orders) : s_(s), work_(work) {
    // The Frags of all contigs go straight into the COMPACT wire form (floria_pileup_packed: a presence bit per SNP of a read's span, a 2-bit allele and a
    // quality byte per call — a fifth of the CSR bytes on the PCIe link), in ONE pinned buffer laid out field by field at the offsets
    // floria_hip_pack_pileups_batch uses, so that every field of a chunk of contigs is one DMA.  Alleles beyond the 2-bit envelope are refused here.
    const size_t n = work.size();
    std::vector<uint64_t> rp(n + 1, 0), cp(n + 1, 0), pb(n + 1, 0);
    for (size_t i = 0; i < n; ++i) {
        uint64_t C = 0, bits = 0;
        for (const Frag& f : work[i].all_frags) {
            if (f.last_position < f.first_position) throw Error(FLORIA_E_INVALID, "Frag with last_position < first_position");
            C += f.seq_dict.size(); bits += (uint64_t)f.last_position - f.first_position + 1;
        }
        if (bits >= (1ull << 32) || C >= (1ull << 32)) throw Error(FLORIA_E_UNSUPPORTED, "contig too large for one pileup (2^32 cells / span bits)");
        rp[i + 1] = rp[i] + work[i].all_frags.size(); cp[i + 1] = cp[i] + C; pb[i + 1] = pb[i] + (bits + 7) / 8 + 1;
    }
    const uint64_t R = rp[n], C = cp[n];
    // set orders: only for the contigs that hold a fragment whose position set is not that of one CIGAR walk (every other contig's are emulated on the device)
    std::vector<char> needs_order(n, 0);
    uint64_t order_cells = 0;
    if (with_set_orders)
        for (size_t i = 0; i < n; ++i) {
            for (const Frag& f : work[i].all_frags) if (f.other_set_order()) { needs_order[i] = 1; break; }
            if (needs_order[i]) order_cells += cp[i + 1] - cp[i];
        }
    size_t cur = 0;
    auto seg = [&](uint64_t bytes) { const size_t o = cur; cur += (size_t)((bytes + 63) & ~(uint64_t)63); return o; };
    const size_t o_ro = seg(4 * (R + n)), o_bo = seg(4 * (R + n)), o_fi = seg(4 * R), o_la = seg(4 * R), o_pr = seg(pb[n] + 16), o_a2 = seg(C / 4 + n + 16), o_qu = seg(C + 16),
                 o_so = seg(4 * order_cells + 16);
    uint64_t so_at = 0;
    pinned_ = floria_hip_host_alloc(cur + 64);
    if (!pinned_) throw Error(FLORIA_E_NOMEM, floria_hip_last_error());
    char* B = (char*)pinned_;
    memset(B + o_pr, 0, pb[n] + 16); memset(B + o_a2, 0, C / 4 + n + 16);
    piles_.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const std::vector<Frag>& fr = work[i].all_frags;
        uint32_t* ro = (uint32_t*)(B + o_ro + 4 * (rp[i] + i)); uint32_t* bo = (uint32_t*)(B + o_bo + 4 * (rp[i] + i));
        uint32_t* fi = (uint32_t*)(B + o_fi + 4 * rp[i]);       uint32_t* la = (uint32_t*)(B + o_la + 4 * rp[i]);
        uint8_t* pr = (uint8_t*)(B + o_pr + pb[i]); uint8_t* a2 = (uint8_t*)(B + o_a2 + cp[i] / 4 + i); uint8_t* qu = (uint8_t*)(B + o_qu + cp[i]);
        uint32_t c = 0;
        uint64_t bit = 0;
        for (size_t k = 0; k < fr.size(); ++k) {
            const Frag& f = fr[k];
            if (f.counter_id != k) throw Error(FLORIA_E_INVALID, "all_frags must be sorted with counter_id == index (floria.rs:289-293)");
            ro[k] = c; bo[k] = (uint32_t)bit; fi[k] = f.first_position; la[k] = f.last_position;
            for (const auto& kv : f.seq_dict) {                                   // (ascending SNP positions: a sorted map)
                if (kv.first < f.first_position || kv.first > f.last_position) throw Error(FLORIA_E_INVALID, "Frag with a call outside [first_position, last_position]");
                if (kv.second > 3) throw Error(FLORIA_E_UNSUPPORTED, "allele index > 3");
                const uint64_t bi = bit + (kv.first - f.first_position);
                pr[bi >> 3] |= (uint8_t)(1u << (bi & 7));
                a2[c >> 2] |= (uint8_t)(kv.second << (2 * (c & 3)));
                qu[c] = f.qual_dict.at(kv.first);
                ++c;
            }
            bit += (uint64_t)f.last_position - f.first_position + 1;
        }
        ro[fr.size()] = c; bo[fr.size()] = (uint32_t)bit;
        uint32_t* so = nullptr;
        if (needs_order[i]) {
            so = (uint32_t*)(B + o_so) + so_at; so_at += c;
            for (size_t k = 0; k < fr.size(); ++k) {
                const Frag& f = fr[k];
                const std::vector<SnpPosition> ord = f.positions_order();
                if (ord.size() != f.seq_dict.size()) throw Error(FLORIA_E_INVALID, "Frag whose position provenance does not match its calls");
                for (size_t j = 0; j < ord.size(); ++j) {
                    const auto it = f.seq_dict.find(ord[j]);
                    if (it == f.seq_dict.end()) throw Error(FLORIA_E_INVALID, "Frag whose position provenance does not match its calls");
                    so[ro[k] + j] = (uint32_t)(it - f.seq_dict.begin());
                }
            }
        }
        piles_[i] = floria_pileup_packed{ro, fi, la, bo, pr, a2, qu, (uint32_t)fr.size(), so};
    }
    handles_.assign(n, nullptr);
}
Batch::~Batch() {
    for (floria_hip_contig* h : handles_) if (h) floria_hip_contig_free(h);
    if (pinned_) floria_hip_host_free(pinned_);
}

void Batch::generate_hap_graphs(const Options& o) {
    std::vector<uint32_t> bc, bs, be, b0;
    for (size_t i = 0; i < work_.size(); ++i) {
        ContigWork& w = work_[i];
        w.iter_vec = get_range_with_lengths(*w.snp_to_genome_pos, o.block_length, o.block_length / 3, o.snp_density);
        b0.push_back((uint32_t)bs.size());
        for (auto& r : w.iter_vec) { bc.push_back((uint32_t)i); bs.push_back(r.first); be.push_back(r.second); }
    }
    b0.push_back((uint32_t)bs.size());
    floria_params prm{o.epsilon, (uint32_t)o.max_ploidy, (uint32_t)o.max_number_solns, o.ploidy_sensitivity, o.stopping_heuristic ? 1 : 0};
    floria_block_result* res = nullptr;
    check(floria_hip_phase_pileups_batch_packed(s_.ctx(), piles_.data(), (uint32_t)piles_.size(), bc.data(), bs.data(), be.data(), (uint32_t)bs.size(), &prm, &res, handles_.data()));
    floria_hap_graph* hg = nullptr;
    int rc = floria_hip_hap_graph(s_.ctx(), res, &hg);
    if (rc) { floria_hip_block_result_free(res); check(rc); }
    for (size_t i = 0; i < work_.size(); ++i) work_[i].hap_graph = build_columns(res, hg, b0[i], b0[i + 1], work_[i].all_frags, work_[i].iter_vec);
    floria_hip_hap_graph_free(hg);
    floria_hip_block_result_free(res);
}

namespace {
struct GroupCsr { std::vector<uint32_t> gc, reads, rng; std::vector<uint64_t> off{0}; };
GroupCsr groups_of(const std::vector<ContigWork>& work, bool final_parts) {
    GroupCsr g;
    for (size_t i = 0; i < work.size(); ++i) {
        const auto& parts = final_parts ? work[i].final_parts : work[i].path_parts;
        const auto& ranges = final_parts ? work[i].final_ranges : work[i].path_ranges;
        for (size_t k = 0; k < parts.size(); ++k) {
            for (const Frag* f : parts[k]) g.reads.push_back((uint32_t)f->counter_id);
            g.off.push_back(g.reads.size()); g.gc.push_back((uint32_t)i);
            g.rng.push_back(ranges[k].first); g.rng.push_back(ranges[k].second);
        }
    }
    return g;
}
}  // namespace

void Batch::process_reads_for_final_parts(const Options& o) {
    if (o.reassign_short) throw Error(FLORIA_E_UNSUPPORTED, "--reassign-short (hidden flag) is not supported");
    const GroupCsr g = groups_of(work_, false);
    floria_groups** out = nullptr;
    check(floria_hip_reassign_batch(s_.ctx(), handles_.data(), (uint32_t)handles_.size(), g.gc.data(), g.off.data(), g.reads.data(), g.rng.data(), (uint32_t)g.gc.size(),
                                    nullptr, nullptr, o.epsilon, &out));
    for (size_t i = 0; i < work_.size(); ++i) {
        ContigWork& w = work_[i];
        const floria_groups* G = out[i];
        w.final_parts.assign(G->n_groups, {}); w.final_ranges.assign(G->n_groups, {});
        for (uint32_t k = 0; k < G->n_groups; ++k) {
            for (uint64_t x = G->grp_off[k]; x < G->grp_off[k + 1]; ++x) w.final_parts[k].push_back(&w.all_frags[G->grp_read[x]]);
            w.final_ranges[k] = {G->range[2 * k], G->range[2 * k + 1]};
        }
    }
    floria_hip_groups_array_free(out, (uint32_t)handles_.size());
}

void Batch::stats_and_hapq(const Options& o) {
    const GroupCsr g = groups_of(work_, true);
    const uint32_t ng = (uint32_t)g.gc.size();
    std::vector<double> st(4 * (size_t)ng + 4, 0.0), rel(ng + 1, 0.0), avg(work_.size() + 1, 0.0);
    std::vector<uint8_t> hq(ng + 1, 0);
    std::vector<std::vector<uint64_t>> pos(work_.size());
    std::vector<const uint64_t*> pp(work_.size());
    std::vector<uint32_t> ns(work_.size());
    for (size_t i = 0; i < work_.size(); ++i) { pos[i].assign(work_[i].snp_to_genome_pos->begin(), work_[i].snp_to_genome_pos->end()); pp[i] = pos[i].data(); ns[i] = (uint32_t)pos[i].size(); }
    if (ng) check(floria_hip_haploset_stats(s_.ctx(), handles_.data(), (uint32_t)handles_.size(), g.gc.data(), g.off.data(), g.reads.data(), g.rng.data(), ng, st.data()));
    check(floria_hip_hapq_batch(s_.ctx(), handles_.data(), (uint32_t)handles_.size(), g.gc.data(), g.off.data(), g.reads.data(), g.rng.data(), ng, pp.data(), ns.data(),
                                (uint64_t)o.block_length, hq.data(), rel.data(), avg.data()));
    size_t k = 0;
    for (size_t i = 0; i < work_.size(); ++i) {
        ContigWork& w = work_[i];
        const size_t m = w.final_parts.size();
        w.stats.assign(st.begin() + 4 * k, st.begin() + 4 * (k + m));
        w.hq.hapqs.assign(hq.begin() + k, hq.begin() + k + m); w.hq.rel_err.assign(rel.begin() + k, rel.begin() + k + m); w.hq.avg_err = avg[i];
        k += m;
    }
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

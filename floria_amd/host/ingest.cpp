// ingest.cpp — BAM / VCF / FASTA -> Frags (file_reader.rs), from scratch on zlib: the reference links htslib, which this image
// does not have.  What is restated:
//   get_contigs_to_phase           file_reader.rs:738-746   BAM header target names
//   get_vcf_profile (+ get_genotypes_from_vcf_hts)   :113-175, :239-314   SNP filter: every allele one character of ACGT (any case)
//   alignment_passed_check         :179-235
//   frag_from_record               :661-736   CIGAR walk (rust-htslib aligned_pairs_full), allele = index of the read base in the
//                                             record's allele list, qual = base quality
//   combine_frags                  :491-659   pair merge, supplementary merge with the distance cutoff
//   get_frags_from_bamvcf_rewrite  :343-460   (frags with SNPs, frags without)
//   get_fasta_seqs                 :462-489   whole sequences (the writers need the contig length)
//   l_epsilon_auto_detect          :749-826   every 1000th pileup column: minority / majority base ratio, read-length quantile
//   alignment::realign             alignment.rs:7-64   around every called SNP the read's 32 bases are globally aligned to the reference's 32
//                                  bases with each allele in turn (match +1, mismatch -1, gap open -2, extend -1) and the best-scoring
//                                  allele replaces the call.  The reference computes the scores with block-aligner 0.4.0 (a third-party
//                                  adaptive banded SIMD aligner, fixed 8-wide blocks); here they come from the exact affine-gap DP, which
//                                  the banded heuristic approximates — identical wherever its band contains an optimal path (parity of this
//                                  step is unpinned: third-party arithmetic).  Only runs when a reference FASTA is given, as in the reference.
// The whole BAM is read once and records are bucketed by contig in file order, so no .bai is needed; `count` (the enumerate
// index frag_from_record stores in counter_id, which breaks ties of Frag::cmp in all_frags.sort(), floria.rs:289) is the
// record's index among the contig's records, as with the reference's fetch().
#include "floria_host.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>
#include <thread>
#include <type_traits>
#include <unordered_map>

namespace floria {

namespace {

std::vector<unsigned char> slurp(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw Error(FLORIA_E_INVALID, "cannot open " + path);
    std::vector<unsigned char> buf;
    if (fseek(f, 0, SEEK_END) == 0) { const long n = ftell(f); if (n > 0) buf.reserve((size_t)n); }
    rewind(f);
    unsigned char chunk[1 << 16];
    for (size_t k; (k = fread(chunk, 1, sizeof chunk, f)) > 0;) buf.insert(buf.end(), chunk, chunk + k);
    fclose(f);
    return buf;
}

// one task per index on up to `threads` threads (first exception rethrown on the caller)
template <class F> void parallel_tasks(size_t n, size_t threads, F&& f) {
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

// BGZF members carry their compressed size (extra subfield 'B','C') and end with ISIZE, so their places in the output are known
// before any is inflated: -> false if `in` is not made of BGZF members only (plain gzip: the serial path takes it)
struct BgzfBlock { size_t in_off, in_len, out_off, out_len; };
struct ByteSpan { const unsigned char* p; size_t n; size_t size() const { return n; } const unsigned char& operator[](size_t i) const { return p[i]; } };
bool bgzf_index(const ByteSpan& in, std::vector<BgzfBlock>& blocks) {
    size_t o = 0, out = 0;
    while (o < in.size()) {
        if (o + 18 > in.size() || in[o] != 0x1f || in[o + 1] != 0x8b || in[o + 2] != 8 || !(in[o + 3] & 4)) return false;
        const size_t xlen = in[o + 10] | (in[o + 11] << 8);
        if (o + 12 + xlen > in.size()) return false;
        size_t x = o + 12, bsize = 0;
        while (x + 4 <= o + 12 + xlen) {
            const size_t slen = in[x + 2] | (in[x + 3] << 8);
            if (in[x] == 'B' && in[x + 1] == 'C' && slen == 2 && x + 6 <= o + 12 + xlen) bsize = (size_t)(in[x + 4] | (in[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || o + bsize > in.size()) return false;
        uint32_t isize; memcpy(&isize, &in[o + bsize - 4], 4);
        blocks.push_back({o, bsize, out, isize});
        o += bsize; out += isize;
    }
    return true;
}

// BGZF = concatenated gzip members; inflate them all
std::vector<unsigned char> bgzf_inflate_all(const std::vector<unsigned char>& in, const std::string& what, size_t threads) {
    std::vector<unsigned char> out;
    std::vector<BgzfBlock> blocks;
    if (threads > 1 && bgzf_index(ByteSpan{in.data(), in.size()}, blocks)) {
        out.resize(blocks.empty() ? 0 : blocks.back().out_off + blocks.back().out_len);
        const size_t per = 64;                                                     // members per task
        parallel_tasks((blocks.size() + per - 1) / per, threads, [&](size_t t) {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) throw Error(FLORIA_E_NOMEM, "inflateInit2 failed");
            for (size_t b = t * per; b < std::min(blocks.size(), (t + 1) * per); ++b) {
                const BgzfBlock& k = blocks[b];
                zs.next_in = const_cast<unsigned char*>(in.data()) + k.in_off; zs.avail_in = (uInt)k.in_len;
                unsigned char dummy;
                zs.next_out = k.out_len ? out.data() + k.out_off : &dummy; zs.avail_out = (uInt)k.out_len;
                const int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.avail_out != 0) { inflateEnd(&zs); throw Error(FLORIA_E_INVALID, what + " is not a valid BGZF/gzip file"); }
                inflateReset(&zs);
            }
            inflateEnd(&zs);
        });
        return out;
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) throw Error(FLORIA_E_NOMEM, "inflateInit2 failed");
    zs.next_in = const_cast<unsigned char*>(in.data());
    zs.avail_in = (uInt)std::min<size_t>(in.size(), 0x7fffffffu);
    size_t consumed_total = 0;
    std::vector<unsigned char> buf(1 << 20);
    while (consumed_total < in.size()) {
        zs.next_out = buf.data(); zs.avail_out = (uInt)buf.size();
        const unsigned char* before = zs.next_in;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        consumed_total += (size_t)(zs.next_in - before);
        out.insert(out.end(), buf.data(), buf.data() + (buf.size() - zs.avail_out));
        if (rc == Z_STREAM_END) {
            if (consumed_total >= in.size()) break;
            inflateReset(&zs);
            zs.next_in = const_cast<unsigned char*>(in.data()) + consumed_total;
            zs.avail_in = (uInt)std::min<size_t>(in.size() - consumed_total, 0x7fffffffu);
        } else if (rc != Z_OK) { inflateEnd(&zs); throw Error(FLORIA_E_INVALID, what + " is not a valid BGZF/gzip file"); }
        else if (zs.avail_in == 0 && consumed_total < in.size()) {
            zs.next_in = const_cast<unsigned char*>(in.data()) + consumed_total;
            zs.avail_in = (uInt)std::min<size_t>(in.size() - consumed_total, 0x7fffffffu);
        }
    }
    inflateEnd(&zs);
    return out;
}

struct Cursor {
    const unsigned char* p; size_t n, o = 0;
    const std::string& what;
    void need(size_t k) const { if (o + k > n) throw Error(FLORIA_E_INVALID, what + " is truncated"); }
    uint32_t u32() { need(4); uint32_t v; memcpy(&v, p + o, 4); o += 4; return v; }
    int32_t i32() { return (int32_t)u32(); }
    uint16_t u16() { need(2); uint16_t v; memcpy(&v, p + o, 2); o += 2; return v; }
    uint8_t u8() { need(1); return p[o++]; }
};

constexpr uint16_t F_PAIRED1 = 64, F_PAIRED2 = 128, F_SECONDARY = 256, F_SUPP = 2048, F_ERRORS = 1796;

// alignment_passed_check (file_reader.rs:179-235) -> (passed, is_supp)
std::pair<bool, bool> alignment_passed_check(uint16_t flags, uint8_t mapq, bool use_supplementary, bool filter_supplementary, uint8_t mapq_cutoff) {
    const bool is_paired = (flags & F_PAIRED1) || (flags & F_PAIRED2);
    bool is_supp = false;
    if (flags & F_SUPP) {
        is_supp = true;
        if (is_paired) return {false, true};
        if (!use_supplementary) return {false, true};
        if (filter_supplementary && mapq < 60) return {false, true};
    }
    if (mapq < mapq_cutoff) return {false, is_supp};
    if (flags & F_ERRORS) return {false, is_supp};
    if (flags & F_SECONDARY) return {false, is_supp};
    return {true, is_supp};
}

inline int cig_op(uint32_t c) { return (int)(c & 15); }
inline uint32_t cig_len(uint32_t c) { return c >> 4; }
// M I D N S H P = X  ->  consumes query / reference
inline bool consumes_q(int op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }
inline bool consumes_r(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }

int64_t reference_end(const BamRecord& r) {                 // bam_endpos
    int64_t len = 0;
    for (size_t k = 0; k < r.cigar.size(); ++k) { const uint32_t c = r.cigar[k]; if (consumes_r(cig_op(c))) len += cig_len(c); }
    return (int64_t)r.pos + (len ? len : 1);
}

// frag_from_record (file_reader.rs:661-736)
Frag frag_from_record(const BamRecord& rec, const std::map<GnPosition, SnpPosition>& snp_positions, const std::map<GnPosition, std::vector<Genotype>>& pos_allele_map, size_t counter_id,
                      bool keep_sequences) {
    Frag frag;
    frag.id = std::string(rec.qname); frag.counter_id = counter_id;
    frag.is_paired = (rec.flags & F_PAIRED1) || (rec.flags & F_PAIRED2);
    frag.first_position = UINT32_MAX; frag.last_position = 0;
    size_t leading_hardclips = 0;
    if ((rec.flags & F_SUPP) && !rec.cigar.empty() && cig_op(rec.cigar[0]) == 5) leading_hardclips = cig_len(rec.cigar[0]);
    frag.first_pos_base = (GnPosition)rec.pos;
    frag.last_pos_base = (GnPosition)reference_end(rec);
    size_t q = 0;
    int64_t r = rec.pos;
    // the SNPs of the contig the alignment spans: walk them together with the CIGAR (positions ascend in both)
    auto snp_it = snp_positions.lower_bound((GnPosition)std::max<int64_t>(0, r));
    for (size_t ck = 0; ck < rec.cigar.size(); ++ck) {
        const uint32_t c = rec.cigar[ck];
        const int op = cig_op(c);
        const uint32_t len = cig_len(c);
        if (op == 0 || op == 7 || op == 8) {                                   // aligned pairs (Some(q), Some(r))
            while (snp_it != snp_positions.end() && (int64_t)snp_it->first < r) ++snp_it;
            while (snp_it != snp_positions.end() && (int64_t)snp_it->first < r + (int64_t)len) {
                const GnPosition genome_pos = snp_it->first;
                const size_t seq_pos = q + (size_t)((int64_t)genome_pos - r);
                if (seq_pos < rec.seq.size()) {
                    const Genotype readbase = (Genotype)rec.seq[seq_pos];
                    const auto& alleles = pos_allele_map.at(genome_pos);
                    for (size_t i = 0; i < alleles.size(); ++i)
                        if (readbase == alleles[i]) {
                            const SnpPosition snp_pos = snp_it->second;
                            frag.seq_dict[snp_pos] = (Genotype)i;                  // (a later alignment position of the same SNP overwrites, as insert does)
                            frag.qual_dict[snp_pos] = seq_pos < rec.qual.size() ? rec.qual[seq_pos] : 255;
                            if (snp_pos < frag.first_position) frag.first_position = snp_pos;
                            if (snp_pos > frag.last_position) frag.last_position = snp_pos;
                            frag.snp_pos_to_seq_pos[snp_pos] = {0, seq_pos + leading_hardclips};
                            break;
                        }
                }
                ++snp_it;
            }
        }
        if (consumes_q(op)) q += len;
        if (consumes_r(op)) r += len;                                          // D / N over a SNP: pair[0] is None -> no call
    }
    frag.seq_len[0] = rec.seq.size();
    if (keep_sequences) {                                                      // :728-734
        frag.seq_string[0].resize(rec.seq.size());
        for (size_t i = 0; i < rec.seq.size(); ++i) {
            const char c = rec.seq[i];
            frag.seq_string[0][i] = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : ((c == 'a' || c == 'c' || c == 'g' || c == 't') ? (char)(c - 32) : 'A');
        }
        frag.qual_string[0].resize(rec.qual.size());
        for (size_t i = 0; i < rec.qual.size(); ++i) frag.qual_string[0][i] = rec.qual[i] > 222 ? 255 : (uint8_t)(rec.qual[i] + 33);     // checked_add(33).unwrap_or(255)
    }
    return frag;
}

struct Tagged { uint16_t flags; Frag frag; };

// global alignment score of two byte strings, affine gaps: a gap of length n costs open + (n - 1) * extend
int nw_affine_score(const unsigned char* q, int nq, const unsigned char* r, int nr) {
    constexpr int MATCH = 1, MISMATCH = -1, OPEN = -2, EXTEND = -1, NEG = -(1 << 28);
    static thread_local std::vector<int> Mv, Iv, Dv;             // (M: ends in a pair, I: gap in r, D: gap in q), rolling rows
    Mv.assign(nr + 1, NEG); Iv.assign(nr + 1, NEG); Dv.assign(nr + 1, NEG);
    Mv[0] = 0;
    for (int j = 1; j <= nr; ++j) Dv[j] = OPEN + (j - 1) * EXTEND;
    for (int i = 1; i <= nq; ++i) {
        int m_diag = Mv[0], i_diag = Iv[0], d_diag = Dv[0];
        Mv[0] = NEG; Dv[0] = NEG; Iv[0] = OPEN + (i - 1) * EXTEND;
        for (int j = 1; j <= nr; ++j) {
            const int m_up = Mv[j], i_up = Iv[j], d_up = Dv[j];
            const int best_diag = std::max(m_diag, std::max(i_diag, d_diag));
            const auto up = [](unsigned char c) { return (unsigned char)(c >= 'a' && c <= 'z' ? c - 32 : c); };
            const int sub = up(q[i - 1]) == up(r[j - 1]) ? MATCH : MISMATCH;
            const int m_new = best_diag + sub;
            const int i_new = std::max(std::max(m_up, d_up) + OPEN, i_up + EXTEND);               // consume q[i-1] against a gap
            const int d_new = std::max(std::max(Mv[j - 1], Iv[j - 1]) + OPEN, Dv[j - 1] + EXTEND);   // consume r[j-1] against a gap
            m_diag = m_up; i_diag = i_up; d_diag = d_up;
            Mv[j] = m_new; Iv[j] = i_new; Dv[j] = d_new;
        }
    }
    return std::max(Mv[nr], std::max(Iv[nr], Dv[nr]));
}

// alignment::realign (alignment.rs:7-64)
// `queue`: calls the exact shortcut cannot decide are not scored here but appended to the queue (windows + where the result goes) for
// floria_hip_realign, the same DP on the device; null = score them on the host (--ingest-only, tests)
void realign(const std::string& ref_gn, Frag& frag, const BamSeqView& read_seq, const std::map<SnpPosition, GnPosition>& var_to_gn_pos,
             const std::map<GnPosition, std::vector<Genotype>>& gn_pos_to_allele, RealignQueue* queue) {
    constexpr size_t flank = 16;
    for (auto& kv : frag.seq_dict) {
        const size_t snp_gn_pos = var_to_gn_pos.at(kv.first);
        const size_t snp_q_pos = frag.snp_pos_to_seq_pos.at(kv.first).second;
        if (flank > snp_gn_pos || flank + snp_gn_pos >= ref_gn.size() || flank > snp_q_pos || flank + snp_q_pos >= read_seq.size()) continue;
        unsigned char q[2 * flank], r[2 * flank];
        for (size_t i = 0; i < 2 * flank; ++i) {
            const char c = read_seq[snp_q_pos - flank + i];                      // DnaString::from_acgt_bytes: anything but ACGT becomes A
            q[i] = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? (unsigned char)c : ((c == 'a' || c == 'c' || c == 'g' || c == 't') ? (unsigned char)(c - 32) : 'A');
            r[i] = (unsigned char)ref_gn[snp_gn_pos - flank + i];
        }
        const auto& alleles = gn_pos_to_allele.at(snp_gn_pos);
        // Exact shortcut.  Both windows have 2 * flank = 32 bytes, so an alignment with gaps has g >= 1 gap columns on EACH side and
        // scores at most (32 - g) + 2 * (OPEN + (g - 1) * EXTEND) = 30 - 3g <= 27.  With h mismatches outside the SNP column the
        // ungapped alignment scores 32 - 2h for an allele equal to the read's base and 30 - 2h for any other.  So for h <= 2 the
        // first allele equal to the read's base wins strictly (28 > 27); with no such allele and h <= 1 every allele scores
        // 30 - 2h >= 28 and the first one is kept (`score > best_score`).  Everything else takes the DP.
        {
            const auto up = [](unsigned char c) { return (unsigned char)(c >= 'a' && c <= 'z' ? c - 32 : c); };
            int h = 0;
            for (size_t i = 0; i < 2 * flank && h <= 2; ++i) h += (i != flank && q[i] != up(r[i]));
            if (h <= 2) {
                size_t a = 0;
                while (a < alleles.size() && up(alleles[a]) != q[flank]) ++a;
                if (a < alleles.size()) { kv.second = (Genotype)a; continue; }
                if (h <= 1) { kv.second = 0; continue; }
            }
        }
        if (queue && alleles.size() <= FLORIA_MAX_ALLELES) {
            const auto up = [](unsigned char c) { return (unsigned char)(c >= 'a' && c <= 'z' ? c - 32 : c); };
            for (size_t i = 0; i < 2 * flank; ++i) { queue->read_windows.push_back(q[i]); queue->ref_windows.push_back(up(r[i])); }
            for (size_t a = 0; a < FLORIA_MAX_ALLELES; ++a) queue->alleles.push_back(a < alleles.size() ? up(alleles[a]) : 0);
            queue->n_alleles.push_back((uint8_t)alleles.size());
            // (the entry lives in the FlatMap's heap storage: it stays where it is when the Frag or the vector holding it MOVES — Frag is nothrow-movable,
            // asserted below — and nothing is inserted into seq_dict between here and realign_queue_on_device; copying a Frag would invalidate it)
            queue->dst.push_back(&kv.second);
            continue;
        }
        int best_score = INT32_MIN;
        Genotype best_geno = 0;
        for (size_t a = 0; a < alleles.size(); ++a) {
            r[flank] = alleles[a];
            const int score = nw_affine_score(q, 2 * flank, r, 2 * flank);
            if (score > best_score) { best_score = score; best_geno = (Genotype)a; }
        }
        kv.second = best_geno;
    }
}

// the records of bam.raw[begin, end) (whole records) -> bam.records (views into bam.raw) and bam.by_tid
void decode_records(BamFile& bam, size_t begin, size_t end, const std::string& path, size_t threads) {
    // record boundaries first (a walk over the block_size fields), then the records are decoded independently
    const std::vector<unsigned char>& raw = bam.raw;
    Cursor c{raw.data(), end, begin, path};
    std::vector<size_t> rec_off;
    while (c.o < end) {
        const uint32_t block_size = c.u32(); c.need(block_size);
        rec_off.push_back(c.o);
        c.o += block_size;
    }
    rec_off.push_back(end + 4);
    bam.records.resize(rec_off.size() - 1);
    const size_t per = 256;
    parallel_tasks((bam.records.size() + per - 1) / per, threads, [&](size_t t) {
        for (size_t i = t * per; i < std::min(bam.records.size(), (t + 1) * per); ++i) {
            Cursor d{raw.data(), rec_off[i + 1] - 4, rec_off[i], path};             // (bounded by the record's own end)
            BamRecord& r = bam.records[i];
            r.tid = d.i32(); r.pos = d.i32();
            const uint8_t l_read_name = d.u8(); r.mapq = d.u8(); (void)d.u16();
            const uint16_t n_cigar = d.u16(); r.flags = d.u16();
            const uint32_t l_seq = d.u32(); (void)d.i32(); (void)d.i32(); (void)d.i32();
            d.need(l_read_name); r.qname = std::string_view((const char*)raw.data() + d.o, l_read_name ? l_read_name - 1 : 0); d.o += l_read_name;
            d.need((size_t)4 * n_cigar); r.cigar = BamCigarView{raw.data() + d.o, n_cigar}; d.o += (size_t)4 * n_cigar;
            d.need((size_t)(l_seq + 1) / 2 + l_seq);
            r.seq = BamSeqView{raw.data() + d.o, l_seq}; d.o += (l_seq + 1) / 2;
            r.qual = BamBytesView{raw.data() + d.o, l_seq}; d.o += l_seq;
            // An alignment with more than 65 535 CIGAR operations keeps its real CIGAR in the CG:B,I tag and a placeholder `<l_seq>S<ref_len>N` in the
            // CIGAR field (SAM spec 4.2.2); htslib's bam_read1 puts the real one back (bam_tag2cigar), so the reference sees it.  The other tags are skipped.
            if (n_cigar == 2 && (r.cigar[0] & 15u) == 4u && (r.cigar[0] >> 4) == l_seq && (r.cigar[1] & 15u) == 3u) {
                const size_t end = rec_off[i + 1] - 4;
                while (d.o + 3 <= end) {
                    const unsigned char t0c = raw[d.o], t1c = raw[d.o + 1], ty = raw[d.o + 2];
                    d.o += 3;
                    size_t len = 0;
                    if (ty == 'A' || ty == 'c' || ty == 'C') len = 1;
                    else if (ty == 's' || ty == 'S') len = 2;
                    else if (ty == 'i' || ty == 'I' || ty == 'f') len = 4;
                    else if (ty == 'Z' || ty == 'H') { while (d.o + len < end && raw[d.o + len]) ++len; ++len; }
                    else if (ty == 'B') {
                        d.need(5);
                        const unsigned char sub = raw[d.o];
                        uint32_t cnt; memcpy(&cnt, raw.data() + d.o + 1, 4);
                        const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                        if (t0c == 'C' && t1c == 'G' && sub == 'I') { d.o += 5; d.need((size_t)4 * cnt); r.cigar = BamCigarView{raw.data() + d.o, cnt}; break; }
                        len = 5 + es * cnt;
                    } else break;                                                    // unknown type: stop looking (the placeholder CIGAR stays)
                    if (d.o + len > end) break;
                    d.o += len;
                }
            }
        }
    });
    for (size_t i = 0; i < bam.records.size(); ++i) {
        const int32_t tid = bam.records[i].tid;
        if (tid >= 0 && (size_t)tid < bam.by_tid.size()) bam.by_tid[tid].push_back((uint32_t)i);
    }
}

}  // namespace

BamFile read_bam(const std::string& path, size_t threads) {
    const bool trace = getenv("FLORIA_HOST_TRACE") != nullptr;
    const auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    const std::vector<unsigned char> packed = slurp(path);
    if (trace) fprintf(stderr, "[read_bam] file %.3fs (%zu MiB)\n", now() - t0, packed.size() >> 20);
    t0 = now();
    BamFile bam;
    bam.raw = bgzf_inflate_all(packed, path, threads);
    const std::vector<unsigned char>& raw = bam.raw;
    if (trace) fprintf(stderr, "[read_bam] inflate %.3fs (%zu MiB)\n", now() - t0, raw.size() >> 20);
    t0 = now();
    Cursor c{raw.data(), raw.size(), 0, path};
    c.need(4);
    if (memcmp(raw.data(), "BAM\1", 4) != 0) throw Error(FLORIA_E_INVALID, path + " is not a BAM file");
    c.o = 4;
    const uint32_t l_text = c.u32(); c.need(l_text); c.o += l_text;
    const uint32_t n_ref = c.u32();
    for (uint32_t i = 0; i < n_ref; ++i) {
        const uint32_t l_name = c.u32(); c.need(l_name);
        bam.target_names.emplace_back((const char*)raw.data() + c.o, l_name ? l_name - 1 : 0); c.o += l_name;
        bam.target_len.push_back(c.u32());
    }
    bam.by_tid.resize(bam.target_names.size());
    decode_records(bam, c.o, raw.size(), path, threads);
    if (trace) fprintf(stderr, "[read_bam] records %.3fs\n", now() - t0);
    return bam;
}

static_assert(std::is_nothrow_move_constructible<Frag>::value && std::is_nothrow_move_assignable<Frag>::value,
              "RealignQueue::dst points into Frag::seq_dict: a vector of Frags must MOVE its elements when it grows");

// ---- BamStream -------------------------------------------------------------------------------------------------------------------------
struct BamStream::Impl {
    std::string path;
    size_t threads = 1;
    int fd = -1;
    const unsigned char* map = nullptr;
    size_t map_len = 0;
    std::vector<BgzfBlock> blocks;            // BGZF members (empty: not a BGZF file -> `whole`)
    bool bgzf = false;
    BamFile whole;                            // fallback: the file read at once
    bool whole_given = false;
    std::vector<std::string> names;
    std::vector<uint64_t> lens;
    size_t first_block = 0, first_skip = 0;   // where the records begin: member index and offset inside its inflated bytes
    // position
    size_t bi = 0;                            // next member to inflate
    std::vector<unsigned char> carry;         // inflated bytes not handed out yet (starts at a record boundary)
    int32_t done_upto = 0;                    // targets < done_upto have been handed out
    int32_t last_tid_seen = 0;
    bool eof = false;
    size_t peak = 0;

    void inflate_range(size_t b0, size_t b1, unsigned char* dst) const {          // members [b0, b1) -> dst (contiguous)
        const size_t base = blocks[b0].out_off;
        const size_t per = 32;
        parallel_tasks((b1 - b0 + per - 1) / per, threads, [&](size_t t) {
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) throw Error(FLORIA_E_NOMEM, "inflateInit2 failed");
            for (size_t b = b0 + t * per; b < std::min(b1, b0 + (t + 1) * per); ++b) {
                const BgzfBlock& k = blocks[b];
                zs.next_in = const_cast<unsigned char*>(map) + k.in_off; zs.avail_in = (uInt)k.in_len;
                unsigned char dummy;
                zs.next_out = k.out_len ? dst + (k.out_off - base) : &dummy; zs.avail_out = (uInt)k.out_len;
                const int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.avail_out != 0) { inflateEnd(&zs); throw Error(FLORIA_E_INVALID, path + " is not a valid BGZF/gzip file"); }
                inflateReset(&zs);
            }
            inflateEnd(&zs);
        });
    }
};

BamStream::BamStream(const std::string& path, size_t threads) : p_(new Impl) {
    Impl& I = *p_;
    I.path = path; I.threads = std::max<size_t>(1, threads);
    I.fd = open(path.c_str(), O_RDONLY);
    if (I.fd < 0) throw Error(FLORIA_E_INVALID, "cannot open " + path);
    struct stat st;
    if (fstat(I.fd, &st) != 0 || st.st_size <= 0) throw Error(FLORIA_E_INVALID, path + " is empty or unreadable");
    I.map_len = (size_t)st.st_size;
    void* m = mmap(nullptr, I.map_len, PROT_READ, MAP_PRIVATE, I.fd, 0);
    if (m == MAP_FAILED) throw Error(FLORIA_E_INVALID, "cannot map " + path);
    I.map = (const unsigned char*)m;
    I.bgzf = bgzf_index(ByteSpan{I.map, I.map_len}, I.blocks) && !I.blocks.empty();
    if (!I.bgzf) {                                   // plain gzip (or anything else zlib understands): no member index, the file is read whole
        I.whole = read_bam(path, threads);
        I.names = I.whole.target_names; I.lens = I.whole.target_len;
        I.whole.tid_begin = 0; I.whole.tid_end = (int32_t)I.names.size();
        return;
    }
    // the header: inflate members until magic, text and reference list are complete
    std::vector<unsigned char> head;
    size_t b1 = 0;
    auto have = [&](size_t need) {
        while (head.size() < need && b1 < I.blocks.size()) {
            const size_t o = head.size();
            head.resize(o + I.blocks[b1].out_len);
            if (I.blocks[b1].out_len) { std::vector<unsigned char> tmp(I.blocks[b1].out_len); I.inflate_range(b1, b1 + 1, tmp.data()); memcpy(head.data() + o, tmp.data(), tmp.size()); }
            ++b1;
        }
        if (head.size() < need) throw Error(FLORIA_E_INVALID, path + " is truncated");
    };
    have(12);
    if (memcmp(head.data(), "BAM\1", 4) != 0) throw Error(FLORIA_E_INVALID, path + " is not a BAM file");
    uint32_t l_text; memcpy(&l_text, head.data() + 4, 4);
    size_t o = 8 + (size_t)l_text;
    have(o + 4);
    uint32_t n_ref; memcpy(&n_ref, head.data() + o, 4); o += 4;
    for (uint32_t i = 0; i < n_ref; ++i) {
        have(o + 4);
        uint32_t l_name; memcpy(&l_name, head.data() + o, 4); o += 4;
        have(o + l_name + 4);
        I.names.emplace_back((const char*)head.data() + o, l_name ? l_name - 1 : 0); o += l_name;
        uint32_t ln; memcpy(&ln, head.data() + o, 4); o += 4;
        I.lens.push_back(ln);
    }
    // the records start at inflated offset o: find its member
    size_t b = 0;
    while (b < I.blocks.size() && I.blocks[b].out_off + I.blocks[b].out_len <= o) ++b;
    I.first_block = b; I.first_skip = b < I.blocks.size() ? o - I.blocks[b].out_off : 0;
    rewind();
}
BamStream::~BamStream() {
    if (p_->map) munmap(const_cast<unsigned char*>(p_->map), p_->map_len);
    if (p_->fd >= 0) close(p_->fd);
}
const std::vector<std::string>& BamStream::target_names() const { return p_->names; }
size_t BamStream::peak_buffer_bytes() const { return p_->peak; }
void BamStream::rewind() {
    Impl& I = *p_;
    if (!I.bgzf) { if (I.whole_given) { I.whole = read_bam(I.path, I.threads); I.whole.tid_begin = 0; I.whole.tid_end = (int32_t)I.names.size(); } I.whole_given = false; return; }
    I.bi = I.first_block; I.carry.clear(); I.done_upto = 0; I.last_tid_seen = 0; I.eof = false;
    if (I.bgzf && I.first_block < I.blocks.size() && I.first_skip) {       // the rest of the member the header ends in
        std::vector<unsigned char> tmp(I.blocks[I.first_block].out_len);
        I.inflate_range(I.first_block, I.first_block + 1, tmp.data());
        I.carry.assign(tmp.begin() + (ptrdiff_t)I.first_skip, tmp.end());
        I.bi = I.first_block + 1;
    }
}
bool BamStream::next(BamFile& seg, size_t min_bytes) {
    Impl& I = *p_;
    const int32_t n_targets = (int32_t)I.names.size();
    if (!I.bgzf) {
        if (I.whole_given) return false;
        I.whole_given = true;
        seg = std::move(I.whole);             // (a rewind() after this would need the file again: the whole-file fallback is single-pass unless re-read)
        I.whole = BamFile();
        I.peak = std::max(I.peak, seg.raw.size());
        return true;
    }
    if (I.eof && I.done_upto >= n_targets) return false;
    std::vector<unsigned char> buf;
    size_t want = std::max<size_t>(min_bytes, 1 << 16);
    for (;;) {
        // ---- inflate members until the buffer holds `want` bytes (or the file ends)
        size_t b1 = I.bi, add = 0;
        while (b1 < I.blocks.size() && I.carry.size() + buf.size() + add < want) add += I.blocks[b1++].out_len;
        if (buf.empty()) { buf.swap(I.carry); I.carry.clear(); }
        if (b1 > I.bi) {
            const size_t o = buf.size();
            buf.resize(o + add);
            I.inflate_range(I.bi, b1, buf.data() + o);
            I.bi = b1;
        }
        const bool at_end = I.bi >= I.blocks.size();
        // ---- walk the records: [0, cut) = whole records of targets that are certainly complete
        size_t o = 0, last_start_of_tid = 0, whole_end = 0;
        int32_t cur_tid = INT32_MIN, prev = I.last_tid_seen;
        bool saw_unmapped = false;
        while (o + 4 <= buf.size()) {
            uint32_t bs; memcpy(&bs, buf.data() + o, 4);
            if (bs < 32) throw Error(FLORIA_E_INVALID, I.path + ": malformed BAM record");
            if (o + 4 + (size_t)bs > buf.size()) break;
            int32_t tid; memcpy(&tid, buf.data() + o + 4, 4);
            if (tid < 0) { if (!saw_unmapped) { saw_unmapped = true; last_start_of_tid = o; cur_tid = n_targets; } }      // unplaced reads close the file: every target is complete
            else if (saw_unmapped || tid < prev) throw Error(FLORIA_E_INVALID, I.path + " is not sorted by reference sequence (floria needs a coordinate-sorted, indexed BAM; so does this reader)");
            else if (tid != cur_tid) { cur_tid = tid; last_start_of_tid = o; prev = tid; }
            o += 4 + (size_t)bs;
            whole_end = o;
        }
        if (at_end && whole_end != buf.size()) throw Error(FLORIA_E_INVALID, I.path + " is truncated");
        // Unplaced reads (tid = -1) sort behind every target: once one is seen, every target is complete and nothing that follows is of use (the reference fetches
        // per contig through the index and never reads them, file_reader.rs:389-436).  The stream ends there — carrying an unmapped tail larger than the window
        // from call to call would never make progress.
        const bool closing = at_end || saw_unmapped;
        size_t cut; int32_t complete_upto;
        if (closing) { cut = saw_unmapped ? last_start_of_tid : whole_end; complete_upto = n_targets; }
        else { cut = last_start_of_tid; complete_upto = cur_tid == INT32_MIN ? I.done_upto : std::min(cur_tid, n_targets); }      // the target of the last record may continue
        if (!closing && (cut == 0 || complete_upto <= I.done_upto)) { want = std::max(want * 2, buf.size() * 2); continue; }   // one target fills the buffer: read on
        // ---- hand out
        seg = BamFile();
        seg.target_names = I.names; seg.target_len = I.lens;
        seg.by_tid.resize(I.names.size());
        I.carry.assign(buf.begin() + (ptrdiff_t)cut, buf.begin() + (ptrdiff_t)(closing ? cut : buf.size()));
        buf.resize(cut);
        seg.raw = std::move(buf);
        I.peak = std::max(I.peak, seg.raw.size() + I.carry.size());
        decode_records(seg, 0, seg.raw.size(), I.path, I.threads);
        seg.tid_begin = I.done_upto; seg.tid_end = complete_upto;
        I.done_upto = complete_upto;
        I.last_tid_seen = std::max(I.last_tid_seen, std::min(prev, n_targets));
        if (closing) { I.eof = true; I.carry.clear(); I.bi = I.blocks.size(); }
        return true;
    }
}

std::vector<std::string> get_contigs_to_phase(const BamFile& bam) { return bam.target_names; }

// get_vcf_profile + get_genotypes_from_vcf_hts (file_reader.rs:113-175, 239-314).  rust-htslib's bcf::Reader takes text VCF (plain / gzip / bgzip) and binary
// BCF2 alike; so does this reader: the file is read through zlib (a BGZF file is a multi-member gzip stream) and told apart by its first bytes.
namespace {
struct VcfSink {
    VcfProfile vp;
    const std::vector<std::string>& ref_chroms;
    std::string last_chrom;
    bool have_last = false;
    SnpPosition snp_counter = 1;
    explicit VcfSink(const std::vector<std::string>& rc) : ref_chroms(rc) {}
    static bool is_acgt(char ch) { const char u = (char)(ch & ~0x20); return u == 'A' || u == 'C' || u == 'G' || u == 'T'; }
    // one record: CHROM, 1-based POS, REF + ALT alleles
    void add(const std::string& chrom, long pos1, const std::vector<std::string>& alleles) {
        const bool known = std::find(ref_chroms.begin(), ref_chroms.end(), chrom) != ref_chroms.end();
        if (known && (!have_last || last_chrom != chrom)) { snp_counter = 1; last_chrom = chrom; have_last = true; }       // :277-280
        std::vector<Genotype> al_vec;
        for (const std::string& a : alleles) {
            if (a.size() != 1 || !is_acgt(a[0])) return;                                                                   // :290-300 (an empty allele cannot occur)
            al_vec.push_back((Genotype)a[0]);
        }
        const GnPosition pos0 = (GnPosition)(pos1 - 1);                          // htslib's 0-based pos()
        vp.snp_to_genome_pos[chrom].push_back(pos0);                             // get_genotypes_from_vcf_hts: every contig of the VCF
        if (!known) return;
        vp.vcf_snp_pos_to_gn_pos_map[chrom][snp_counter] = pos0;
        vp.vcf_pos_to_snp_counter_map[chrom][pos0] = snp_counter;
        vp.vcf_pos_allele_map[chrom][pos0] = al_vec;
        ++snp_counter;
    }
};

// BCF2 (VCFv4.x specification, section 6): "BCF\2\2", l_text, the VCF header text, then records {l_shared, l_indiv, CHROM (index into the header's contig
// dictionary), POS (0-based), rlen, QUAL, n_info | n_allele << 16, n_sample | n_fmt << 24, ID, alleles ..}: typed values with a descriptor byte
// (length << 4 | type; length 15 = a typed integer with the real length follows; type 7 = characters).  Only CHROM, POS and the alleles are read.
void read_bcf(gzFile f, const std::string& vcf_file, std::vector<unsigned char> head, VcfSink& sink) {
    auto fail = [&](const char* what) -> void { throw Error(FLORIA_E_INVALID, "BCF " + vcf_file + ": " + what); };
    auto need = [&](std::vector<unsigned char>& buf, size_t n) {
        const size_t have = buf.size();
        if (have >= n) return true;
        buf.resize(n);
        size_t got = have;
        while (got < n) { const int r = gzread(f, buf.data() + got, (unsigned)std::min<size_t>(n - got, 1u << 30)); if (r <= 0) break; got += (size_t)r; }
        buf.resize(got);
        return got >= n;
    };
    auto u32 = [](const unsigned char* q) { return (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24; };
    if (!need(head, 9) || memcmp(head.data(), "BCF\2", 4) != 0 || head[4] < 1) fail("not a BCF2 file");
    const uint32_t l_text = u32(head.data() + 5);
    if (!need(head, 9 + (size_t)l_text)) fail("truncated header");
    // the contig dictionary: ##contig=<ID=name,..[,IDX=k]> lines in order (IDX overrides the position)
    std::vector<std::string> contigs;
    {
        const std::string text((const char*)head.data() + 9, l_text);
        size_t at = 0;
        while (at < text.size()) {
            size_t e = text.find('\n', at); if (e == std::string::npos) e = text.size();
            const std::string line = text.substr(at, e - at);
            at = e + 1;
            if (line.rfind("##contig=<", 0) != 0) continue;
            std::string id; long idx = -1;
            size_t q = 10;
            while (q < line.size() && line[q] != '>') {
                size_t eq = line.find('=', q), stop = q;
                if (eq == std::string::npos) break;
                const std::string key = line.substr(q, eq - q);
                size_t v0 = eq + 1, v1;
                if (v0 < line.size() && line[v0] == '"') { v1 = line.find('"', v0 + 1); if (v1 == std::string::npos) v1 = line.size(); stop = v1 + 1; ++v0; }
                else { v1 = line.find_first_of(",>", v0); if (v1 == std::string::npos) v1 = line.size(); stop = v1; }
                const std::string val = line.substr(v0, v1 - v0);
                if (key == "ID") id = val; else if (key == "IDX") idx = strtol(val.c_str(), nullptr, 10);
                q = stop; if (q < line.size() && line[q] == ',') ++q;
            }
            if (id.empty()) continue;
            if (idx < 0) idx = (long)contigs.size();
            if ((size_t)idx >= contigs.size()) contigs.resize((size_t)idx + 1);
            contigs[(size_t)idx] = id;
        }
    }
    std::vector<unsigned char> rec;
    for (;;) {
        rec.clear();
        if (!need(rec, 8)) { if (rec.empty()) break; fail("truncated record header"); }
        const uint32_t l_shared = u32(rec.data()), l_indiv = u32(rec.data() + 4);
        if (l_shared < 24) fail("record shorter than its fixed fields");
        // only the shared part is parsed: it is bounded (a corrupt or hostile length must become a floria Error, not a bad_alloc), the per-sample part is skipped
        // in bounded reads instead of being buffered
        if (l_shared > (64u << 20)) fail("record with a shared part of more than 64 MB");
        rec.clear();
        if (!need(rec, (size_t)l_shared)) fail("truncated record");
        {
            unsigned char skip[1 << 16];
            for (uint64_t left = l_indiv; left != 0;) {
                const int r = gzread(f, skip, (unsigned)std::min<uint64_t>(left, sizeof skip));
                if (r <= 0) fail("truncated record");
                left -= (uint64_t)r;
            }
        }
        const unsigned char* q = rec.data();
        const unsigned char* const end = q + l_shared;
        const int32_t chrom = (int32_t)u32(q), pos0 = (int32_t)u32(q + 4);
        const uint32_t n_allele = u32(q + 16) >> 16;
        q += 24;
        if (chrom < 0 || (size_t)chrom >= contigs.size() || contigs[(size_t)chrom].empty()) fail("CHROM index outside the header's contig dictionary");
        // typed value: -> (type, count), cursor behind the descriptor
        auto typed = [&](uint32_t& type, uint32_t& count) {
            if (q >= end) fail("typed value beyond the record");
            const unsigned d = *q++;
            type = d & 15u; count = d >> 4;
            if (count == 15) {                                        // the real length is a typed integer
                if (q >= end) fail("typed length beyond the record");
                const unsigned t2 = *q++ & 15u;
                const size_t w = t2 == 1 ? 1 : t2 == 2 ? 2 : t2 == 3 ? 4 : 0;
                if (!w || q + w > end) fail("bad typed length");
                count = w == 1 ? q[0] : w == 2 ? (uint32_t)q[0] | (uint32_t)q[1] << 8 : u32(q);
                q += w;
            }
        };
        auto width = [&](uint32_t type) -> size_t { return type == 1 || type == 7 ? 1 : type == 2 ? 2 : type == 3 || type == 5 ? 4 : 0; };
        uint32_t ty, cnt;
        typed(ty, cnt);                                                // ID
        if (cnt && (!width(ty) || q + (size_t)cnt * width(ty) > end)) fail("bad ID");
        q += (size_t)cnt * width(ty);
        std::vector<std::string> alleles;
        for (uint32_t a = 0; a < n_allele; ++a) {
            typed(ty, cnt);
            if (cnt && (ty != 7 || q + cnt > end)) fail("allele is not a string");
            alleles.emplace_back((const char*)q, cnt);
            q += cnt;
        }
        if (alleles.empty()) continue;
        sink.add(contigs[(size_t)chrom], (long)pos0 + 1, alleles);
    }
}
}  // namespace

VcfProfile get_vcf_profile(const std::string& vcf_file, const std::vector<std::string>& ref_chroms) {
    gzFile f = gzopen(vcf_file.c_str(), "rb");                                 // plain text, gzip and bgzip alike
    if (!f) throw Error(FLORIA_E_INVALID, "cannot open VCF " + vcf_file);
    VcfSink sink(ref_chroms);
    std::vector<char> buf(1 << 16);
    {   // BCF2 or text?
        std::vector<unsigned char> head(5);
        const int got = gzread(f, head.data(), 5);
        head.resize(got > 0 ? (size_t)got : 0);
        if (head.size() == 5 && memcmp(head.data(), "BCF\2", 4) == 0) {
            try { read_bcf(f, vcf_file, std::move(head), sink); } catch (...) { gzclose(f); throw; }
            gzclose(f);
            return std::move(sink.vp);
        }
        if (gzrewind(f) != 0) { gzclose(f); throw Error(FLORIA_E_INVALID, "cannot rewind VCF " + vcf_file); }
    }
    std::string line;
    for (;;) {
        line.clear();
        bool got = false;
        while (gzgets(f, buf.data(), (int)buf.size())) { got = true; line += buf.data(); if (!line.empty() && line.back() == '\n') break; }
        if (!got) break;
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        // CHROM POS ID REF ALT ...
        size_t t[5], k = 0, from = 0;
        while (k < 5) { const size_t x = line.find('\t', from); if (x == std::string::npos) break; t[k++] = x; from = x + 1; }
        if (k < 4) continue;
        const std::string chrom = line.substr(0, t[0]);
        const long pos1 = strtol(line.c_str() + t[0] + 1, nullptr, 10);
        const std::string ref = line.substr(t[2] + 1, t[3] - t[2] - 1);
        const std::string alt = line.substr(t[3] + 1, (k == 5 ? t[4] : line.size()) - t[3] - 1);
        std::vector<std::string> alleles{ref};
        if (alt != ".") { size_t a = 0; for (;;) { const size_t x = alt.find(',', a); alleles.push_back(alt.substr(a, x == std::string::npos ? x : x - a)); if (x == std::string::npos) break; a = x + 1; } }
        sink.add(chrom, pos1, alleles);
    }
    gzclose(f);
    return std::move(sink.vp);
}

std::map<std::string, std::string> get_fasta_seqs(const std::string& fasta_file) {
    gzFile f = gzopen(fasta_file.c_str(), "rb");
    if (!f) throw Error(FLORIA_E_INVALID, "Could not read fasta file " + fasta_file);
    std::map<std::string, std::string> out;
    std::string* cur = nullptr;
    std::vector<char> buf(1 << 16);
    while (gzgets(f, buf.data(), (int)buf.size())) {
        size_t n = strlen(buf.data());
        while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) --n;
        if (n && buf[0] == '>') {
            size_t e = 1;
            while (e < n && buf[e] != ' ' && buf[e] != '\t') ++e;
            cur = &out[std::string(buf.data() + 1, e - 1)];
        } else if (cur) cur->append(buf.data(), n);
    }
    gzclose(f);
    return out;
}

struct ContigIngest::Impl {
    std::vector<std::vector<Tagged>> buckets;                                  // read name -> its passing alignments, in record order
    const std::map<SnpPosition, GnPosition>* snp_to_gn = nullptr;
    int64_t supp_aln_dist_cutoff = 0;
};
ContigIngest::~ContigIngest() = default;
ContigIngest::ContigIngest(ContigIngest&&) noexcept = default;
ContigIngest& ContigIngest::operator=(ContigIngest&&) noexcept = default;

ContigIngest::ContigIngest(const BamFile& bam, const VcfProfile& vp, const Options& o, const std::string& contig, const std::string* ref_seq, RealignQueue* queue)
    : p_(new Impl) {
    const bool filter_supplementary = true, use_supplementary = !o.dont_use_supp_aln;
    const auto tid_it = std::find(bam.target_names.begin(), bam.target_names.end(), contig);
    if (tid_it == bam.target_names.end()) return;
    const int32_t tid = (int32_t)(tid_it - bam.target_names.begin());
    const auto& snp_positions = vp.vcf_pos_to_snp_counter_map.at(contig);
    const auto& pos_allele_map = vp.vcf_pos_allele_map.at(contig);
    const auto& snp_to_gn = vp.vcf_snp_pos_to_gn_pos_map.at(contig);
    p_->snp_to_gn = &snp_to_gn; p_->supp_aln_dist_cutoff = o.supp_aln_dist_cutoff;
    // read name -> its passing alignments, in record order (the reference fills the buckets from a parallel loop)
    std::vector<std::string> names;
    std::unordered_map<std::string, size_t> name_ix;
    std::vector<std::vector<Tagged>>& buckets = p_->buckets;
    size_t count = 0;
    for (const uint32_t rec_ix : bam.by_tid[tid]) {                           // the contig's records in file order, as fetch() yields them
        const BamRecord& rec = bam.records[rec_ix];
        const size_t this_count = count++;
        if (!alignment_passed_check(rec.flags, rec.mapq, use_supplementary, filter_supplementary, o.mapq_cutoff).first) continue;
        auto ins = name_ix.emplace(std::string(rec.qname), names.size());
        if (ins.second) { names.emplace_back(rec.qname); buckets.emplace_back(); }
        Frag fr = frag_from_record(rec, snp_positions, pos_allele_map, this_count, o.output_reads);
        if (ref_seq) realign(*ref_seq, fr, rec.seq, snp_to_gn, pos_allele_map, queue);             // :416-423
        buckets[ins.first->second].push_back({rec.flags, std::move(fr)});
    }
}

std::pair<std::vector<Frag>, std::vector<Frag>> ContigIngest::finish() {
    std::vector<std::vector<Tagged>>& buckets = p_->buckets;
    if (buckets.empty()) return {};
    const std::map<SnpPosition, GnPosition>& snp_to_gn = *p_->snp_to_gn;
    struct { int64_t supp_aln_dist_cutoff; } o{p_->supp_aln_dist_cutoff};
    // combine_frags (:491-659)
    std::vector<Frag> ref_frags;
    for (auto& frags : buckets) {
        if (frags.size() == 2 && frags[0].frag.is_paired && frags[1].frag.is_paired) {
            std::sort(frags.begin(), frags.end(), [](const Tagged& a, const Tagged& b) { return a.flags != b.flags ? a.flags < b.flags : a.frag < b.frag; });
            Tagged &first = frags[0], &second = frags[1];
            Frag *ff, *sf;
            if ((first.flags & F_PAIRED1) == F_PAIRED1) { ff = &first.frag; sf = &second.frag; }
            else if ((first.flags & F_PAIRED2) == F_PAIRED2) { ff = &second.frag; sf = &first.frag; }
            else continue;
            if (!sf->seq_dict.empty()) {                                                   // first_frag.positions.extend(sec_frag.positions) (:541): remember who brought what
                ff->merged_positions = true;
                ff->position_segments.emplace_back(); ff->position_segments.emplace_back();
                for (const auto& kv : ff->seq_dict) ff->position_segments[0].push_back(kv.first);
                for (const auto& kv : sf->seq_dict) ff->position_segments[1].push_back(kv.first);
            }
            for (auto& kv : sf->seq_dict) ff->seq_dict[kv.first] = kv.second;              // extend: the mate's call overwrites
            for (auto& kv : sf->qual_dict) ff->qual_dict[kv.first] = kv.second;
            ff->first_position = std::min(ff->first_position, sf->first_position);
            ff->last_position = std::max(ff->last_position, sf->last_position);
            ff->first_pos_base = std::min(ff->first_pos_base, sf->first_pos_base);
            ff->last_pos_base = std::min(ff->last_pos_base, sf->last_pos_base);           // (min, as in the reference, :547)
            ff->seq_len[1] = sf->seq_len[0];
            ff->seq_string[1] = std::move(sf->seq_string[0]); ff->qual_string[1] = std::move(sf->qual_string[0]);   // :551-554
            for (auto& kv : sf->snp_pos_to_seq_pos) ff->snp_pos_to_seq_pos[kv.first] = {1, kv.second.second};
            ref_frags.push_back(std::move(*ff));
        } else if (frags.size() == 1 && (frags[0].flags & F_SUPP) == 0) {
            ref_frags.push_back(std::move(frags[0].frag));
        } else {
            std::vector<std::pair<SnpPosition, SnpPosition>> supp_intervals;
            for (auto& t : frags) if (!t.frag.seq_dict.empty()) supp_intervals.push_back({t.frag.first_position, t.frag.last_position});
            std::sort(supp_intervals.begin(), supp_intervals.end());
            bool take_primary_only = false;
            for (size_t i = 0; i + 1 < supp_intervals.size(); ++i)
                if ((int64_t)snp_to_gn.at(supp_intervals[i + 1].first) - (int64_t)snp_to_gn.at(supp_intervals[i].second) > o.supp_aln_dist_cutoff) { take_primary_only = true; break; }
            int primary = -1;
            for (size_t i = 0; i < frags.size(); ++i) if ((frags[i].flags & F_SUPP) != F_SUPP) primary = (int)i;   // the LAST primary wins (:606-614)
            if (primary < 0) continue;
            if (take_primary_only) { ref_frags.push_back(std::move(frags[primary].frag)); continue; }
            Frag pf = std::move(frags[primary].frag);
            for (size_t i = 0; i < frags.size(); ++i) {
                if ((int)i == primary) continue;
                Frag& fr = frags[i].frag;
                if (!fr.seq_dict.empty()) {                                                  // :639
                    pf.merged_positions = true;
                    if (pf.position_segments.empty()) { pf.position_segments.emplace_back(); for (const auto& kv : pf.seq_dict) pf.position_segments[0].push_back(kv.first); }
                    pf.position_segments.emplace_back();
                    for (const auto& kv : fr.seq_dict) pf.position_segments.back().push_back(kv.first);
                }
                for (auto& kv : fr.seq_dict) pf.seq_dict[kv.first] = kv.second;
                for (auto& kv : fr.qual_dict) pf.qual_dict[kv.first] = kv.second;
                pf.first_position = std::min(pf.first_position, fr.first_position);
                pf.last_position = std::max(pf.last_position, fr.last_position);
                pf.first_pos_base = std::min(pf.first_pos_base, fr.first_pos_base);
                pf.last_pos_base = std::min(pf.last_pos_base, fr.last_pos_base);
                for (auto& kv : fr.snp_pos_to_seq_pos) pf.snp_pos_to_seq_pos[kv.first] = kv.second;
            }
            ref_frags.push_back(std::move(pf));
        }
    }
    std::pair<std::vector<Frag>, std::vector<Frag>> out;
    for (Frag& f : ref_frags) (f.seq_dict.empty() ? out.second : out.first).push_back(std::move(f));
    buckets.clear();
    return out;
}

std::pair<std::vector<Frag>, std::vector<Frag>> get_frags_from_bamvcf_rewrite(const BamFile& bam, const VcfProfile& vp, const Options& o, const std::string& contig,
                                                                               const std::string* ref_seq) {
    return ContigIngest(bam, vp, o, contig, ref_seq, nullptr).finish();
}

void RealignQueue::append(RealignQueue&& o) {
    read_windows.insert(read_windows.end(), o.read_windows.begin(), o.read_windows.end());
    ref_windows.insert(ref_windows.end(), o.ref_windows.begin(), o.ref_windows.end());
    alleles.insert(alleles.end(), o.alleles.begin(), o.alleles.end());
    n_alleles.insert(n_alleles.end(), o.n_alleles.begin(), o.n_alleles.end());
    dst.insert(dst.end(), o.dst.begin(), o.dst.end());
    o = RealignQueue();
}

// file_reader.rs:749-826.  The htslib pileup engine visits every reference position covered by at least one alignment that
// passes its default mask (unmapped | secondary | qc-fail | duplicate are dropped), contig by contig in coordinate order.
struct EpsilonEstimator::Impl {
    size_t count = 0;
    std::vector<double> err_vec;
    std::vector<size_t> read_lengths;
    bool done = false;
};
EpsilonEstimator::EpsilonEstimator() : p_(new Impl) {}
EpsilonEstimator::~EpsilonEstimator() = default;
bool EpsilonEstimator::done() const { return p_->done; }
void EpsilonEstimator::feed(const BamFile& bam) {
    size_t& count = p_->count;
    std::vector<double>& err_vec = p_->err_vec;
    std::vector<size_t>& read_lengths = p_->read_lengths;
    bool& done = p_->done;
    if (done) return;
    struct Aln { const BamRecord* r; int64_t beg, end; };
    std::vector<std::vector<Aln>> by_tid(bam.target_names.size());
    for (const BamRecord& r : bam.records) {
        if (r.tid < 0 || (size_t)r.tid >= by_tid.size() || (r.flags & F_ERRORS) || r.cigar.empty()) continue;
        by_tid[r.tid].push_back({&r, r.pos, reference_end(r)});
    }
    const size_t stop = 1000;
    auto base_at = [](const BamRecord& r, int64_t pos, char* base) -> bool {          // false: deletion / refskip / not aligned here
        size_t q = 0; int64_t ref = r.pos;
        for (size_t ck = 0; ck < r.cigar.size(); ++ck) {
            const uint32_t c = r.cigar[ck];
            const int op = cig_op(c); const int64_t len = cig_len(c);
            if (consumes_r(op) && pos < ref + len) {
                if (!(op == 0 || op == 7 || op == 8)) return false;
                const size_t sp = q + (size_t)(pos - ref);
                if (sp >= r.seq.size()) return false;
                *base = r.seq[sp];
                return true;
            }
            if (consumes_q(op)) q += (size_t)len;
            if (consumes_r(op)) ref += len;
        }
        return false;
    };
    for (auto& alns : by_tid) {
        if (done) break;
        std::stable_sort(alns.begin(), alns.end(), [](const Aln& a, const Aln& b) { return a.beg < b.beg; });
        size_t next = 0;
        std::vector<Aln> active;
        int64_t pos = 0;
        while (!done && (next < alns.size() || !active.empty())) {
            if (active.empty()) pos = std::max(pos, alns[next].beg);
            while (next < alns.size() && alns[next].beg <= pos) active.push_back(alns[next++]);
            active.erase(std::remove_if(active.begin(), active.end(), [&](const Aln& a) { return a.end <= pos; }), active.end());
            if (active.empty()) continue;
            // a pileup column at `pos`
            if (count % 1000 != 0) {
                // skip ahead: columns are consecutive while the active set is non-empty; stop where the set can change
                int64_t lim = INT64_MAX;
                for (const Aln& a : active) lim = std::min(lim, a.end);
                if (next < alns.size()) lim = std::min(lim, alns[next].beg);
                const int64_t cols = std::max<int64_t>(1, lim - pos);
                const int64_t to_sample = 1000 - (int64_t)(count % 1000);
                const int64_t step = std::min(cols, to_sample);
                count += (size_t)step; pos += step;
                continue;
            }
            double most_base = 0., total_c = 0.;
            double base_cnt[256] = {0};
            for (const Aln& a : active) {
                char b;
                if (!base_at(*a.r, pos, &b)) continue;
                if ((a.r->flags & F_ERRORS) || (a.r->flags & F_SECONDARY) || a.r->seq.empty()) continue;
                read_lengths.push_back(a.r->seq.size());
                base_cnt[(unsigned char)b] += 1.;
            }
            for (double v : base_cnt) { if (v > most_base) most_base = v; total_c += v; }
            ++pos;
            if (total_c < 5.) continue;                                      // (count is not advanced: the next column is sampled too)
            err_vec.push_back((total_c - most_base) / most_base);
            if (err_vec.size() >= stop && !read_lengths.empty()) { done = true; break; }
            ++count;
        }
    }
}
std::pair<size_t, double> EpsilonEstimator::result() {
    std::vector<size_t>& read_lengths = p_->read_lengths;
    std::vector<double>& err_vec = p_->err_vec;
    std::sort(read_lengths.begin(), read_lengths.end());
    if (read_lengths.empty()) return {500, 0.01};
    const size_t q_66 = read_lengths[read_lengths.size() * 66 / 100];
    std::sort(err_vec.begin(), err_vec.end());
    const double med66 = err_vec.empty() ? 0.0 : err_vec[err_vec.size() * 66 / 100];
    return {std::max<size_t>(q_66, 500), std::max(med66, 0.01)};             // constants::MINIMUM_BLOCK_SIZE = 500
}

std::pair<size_t, double> l_epsilon_auto_detect(const BamFile& bam) {
    EpsilonEstimator e;
    e.feed(bam);
    return e.result();
}

}  // namespace floria

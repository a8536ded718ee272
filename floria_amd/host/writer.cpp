// writer.cpp — the on-disk formats of floria, the user-visible contract of the drop-in (file_writer.rs).
//   {contig_dir}/{contig}.vartigs         write_haplotypes            file_writer.rs:699-917
//   {contig_dir}/vartig_info.txt          write_fragset_haplotypes    file_writer.rs:308-369
//   {contig_dir}/{contig}.haplosets       write_all_parts_file        file_writer.rs:919-993
//   {contig_dir}/reads_without_snps.tsv   write_nosnp_reads_parts     file_writer.rs:151-165
//   {out_dir}/contig_ploidy_info.tsv      (appended)                  file_writer.rs:883-914, header constants.rs:24
//   {out_dir}/cmd.log                                                  parse_cmd_line.rs:121-126
// COV / ERR come from floria_hip_haploset_stats (get_errors_cov_from_frags, utils_frags.rs:596-655) and HAPQ / REL_ERR from
// floria_hip_hapq (get_hapq, part_block_manip.rs:517-616), both computed on the device; the allele strings of the vartigs are
// counted here on the host (set_to_seq_dict(.., false), utils_frags.rs:160-175).
#include "floria_host.hpp"

#include <zlib.h>

#include <ftw.h>
#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace floria {

namespace {

void check(int rc) { if (rc != 0) throw Error(rc, floria_hip_last_error()); }

// Rust's `{:.N}` of an f64: exact decimal expansion, ties to even — what glibc's printf does — but NaN / inf spell differently
std::string fmt_f64(double v, int prec) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    char buf[64];
    snprintf(buf, sizeof buf, "%.*f", prec, v);
    return buf;
}

void mkdir_p(const std::string& path) {
    std::string cur;
    for (size_t i = 0; i <= path.size(); ++i) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty() && mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) throw Error(FLORIA_E_INVALID, "cannot create directory " + cur + ": " + strerror(errno));
        }
        if (i < path.size()) cur.push_back(path[i]);
    }
}

// iteration order of the reference's inner map FxHashMap<Genotype, GenotypeCount> (fxhash 0.2.1 + hashbrown): ascending allele
// while <= 3 distinct alleles are present (4 buckets), 0, 2, 1, 3 once all four are (8 buckets) — DESIGN.md §6
void allele_order(const uint32_t* cnt, int* order, int* n) {
    int present[4], np = 0;
    for (int a = 0; a < 4; ++a) if (cnt[a]) present[np++] = a;
    if (np == 4) { order[0] = 0; order[1] = 2; order[2] = 1; order[3] = 3; }
    else for (int i = 0; i < np; ++i) order[i] = present[i];
    *n = np;
}

}  // namespace

void write_run_files(const Options& o, int argc, char** argv, const std::string& note) {
    struct stat st;
    if (stat(o.out_dir.c_str(), &st) == 0 && !o.overwrite)
        throw Error(FLORIA_E_INVALID, "Output directory exists; output directory must not be an existing directory. Use --overwrite to overwrite existing directory.");
    mkdir_p(o.out_dir);
    std::ofstream cmd(o.out_dir + "/cmd.log", std::ios::trunc);
    for (int i = 0; i < argc; ++i) cmd << argv[i] << " ";
    if (!note.empty()) cmd << "\n" << note << "\n";              // (the first line is the reference's; what this build decided beyond the command line follows)
    std::ofstream pl(o.out_dir + "/contig_ploidy_info.tsv", std::ios::trunc);                      // constants.rs:24
    pl << "contig\taverage_straincount\twhole_contig_multiplicity\tapproximate_coverage_ignoring_indels\ttotal_vartig_bases_covered\t"
          "average_straincount_min15hapq\taverage_straincount_min30hapq\taverage_straincount_min45hapq\tavg_err\n";
}

namespace {
// the files of one contig; returns its contig_ploidy_info.tsv row (appended by the caller, so that rows keep the contig order when the
// contigs of a batch are written by several threads)
std::string write_contig_files(const std::vector<std::vector<const Frag*>>& part, const std::vector<std::pair<SnpPosition, SnpPosition>>& ranges,
                               const std::string& dir, const std::string& prefix, const std::string& contig, const std::vector<GnPosition>& snp_to_gn,
                               const std::vector<const Frag*>& snpless_frags, size_t contig_len, const HapqResult& hq, const std::vector<double>& st) {
    mkdir_p(dir);
    const size_t n = part.size();
    // ---- write_haplotypes (:699-917) -----------------------------------------------------------------------------------------------
    const size_t S = snp_to_gn.size();
    std::vector<double> cnt_all(S, 0.), cov_all(S, 0.), cnt15(S, 0.), cnt30(S, 0.), cnt45(S, 0.);
    size_t total_bases_covered = 0;
    FILE* vt = fopen((dir + "/" + contig + ".vartigs").c_str(), "w");
    FILE* vi = fopen((dir + "/vartig_info.txt").c_str(), "w");
    FILE* hs = fopen((dir + "/" + prefix + ".haplosets").c_str(), "w");
    if (!vt || !vi || !hs) { if (vt) fclose(vt); if (vi) fclose(vi); if (hs) fclose(hs); throw Error(FLORIA_E_INVALID, "cannot create output files in " + dir); }
    for (size_t i = 0; i < n; ++i) {
        if (part[i].empty()) continue;
        const SnpPosition left = ranges[i].first, right = ranges[i].second;
        if (left > right || left == 0 || right > S) { fclose(vt); fclose(vi); fclose(hs); throw Error(FLORIA_E_INVALID, "haploset SNP range out of order (the reference panics here)"); }
        const GnPosition lg = snp_to_gn[left - 1], rg = snp_to_gn[right - 1];
        total_bases_covered += rg - lg;
        const double cov = st[4 * i], err = st[4 * i + 1];
        const unsigned hap_q = hq.hapqs[i];
        for (SnpPosition p = left; p <= right; ++p) {
            cnt_all[p - 1] += 1.; cov_all[p - 1] += cov;
            if (hap_q >= 15) cnt15[p - 1] += 1.;
            if (hap_q >= 30) cnt30[p - 1] += 1.;
            if (hap_q >= 45) cnt45[p - 1] += 1.;
        }
        const std::string head = ">HAP" + std::to_string(i) + "." + dir + "\tCONTIG:" + contig + "\tSNPRANGE:" + std::to_string(left) + "-" + std::to_string(right) +
                                 "\tBASERANGE:" + std::to_string(lg + 1) + "-" + std::to_string(rg + 1) + "\tCOV:" + fmt_f64(cov, 3) + "\tERR:" + fmt_f64(err, 4) +
                                 "\tHAPQ:" + std::to_string(hap_q) + "\tREL_ERR:" + fmt_f64(hq.rel_err[i], 3) + "\n";
        fputs(head.c_str(), vt);
        // write_fragset_haplotypes (:308-369): per SNP of the range the unit-count allele histogram of the haploset's reads
        std::vector<uint32_t> hist((size_t)(right - left + 1) * 4, 0);
        for (const Frag* f : part[i])
            for (auto it = f->seq_dict.lower_bound(left); it != f->seq_dict.end() && it->first <= right; ++it) hist[(size_t)(it->first - left) * 4 + (it->second & 3)]++;
        fprintf(vi, ">HAP%zu.%s\tSNPRANGE:%u-%u\n", i, dir.c_str(), left, right);
        std::string alleles;
        for (SnpPosition p = left; p <= right; ++p) {
            const uint32_t* c = &hist[(size_t)(p - left) * 4];
            int order[4], na = 0;
            allele_order(c, order, &na);
            fprintf(vi, "%u:%zu\t", p, (size_t)snp_to_gn[p - 1]);
            if (na == 0) { fputs("?\tNA\t\n", vi); alleles.push_back('?'); continue; }          // 15 + 48 = '?'
            int best = order[0];
            for (int k = 1; k < na; ++k) if (c[order[k]] >= c[best]) best = order[k];         // max_by_key: the LAST maximum in iteration order
            fprintf(vi, "%d\t", best);
            alleles.push_back((char)('0' + best));
            for (int k = 0; k < na; ++k) fprintf(vi, "%s%d:%u", k ? "|" : "", order[k], c[order[k]]);
            fputs("\t\n", vi);
        }
        fputs(alleles.c_str(), vt); fputc('\n', vt);
        // write_all_parts_file (:919-993): the same header, then the reads sorted by Frag::cmp (ascending counter_id)
        fputs(head.c_str(), hs);
        std::vector<const Frag*> vec_part(part[i]);
        std::sort(vec_part.begin(), vec_part.end(), [](const Frag* a, const Frag* b) { return *a < *b; });
        for (const Frag* f : vec_part) fprintf(hs, "%s\t%u\t%u\n", f->id.c_str(), f->first_position, f->last_position);
    }
    fclose(vt); fclose(vi); fclose(hs);
    // contig_ploidy_info.tsv (:883-914)
    std::string row;
    {
        size_t num_nonzero = 0;
        double sum_all = 0., sum15 = 0., sum30 = 0., sum45 = 0., sum_cov = 0.;
        for (size_t p = 0; p < S; ++p) { if (cnt_all[p] > 0.) ++num_nonzero; sum_all += cnt_all[p]; sum15 += cnt15[p]; sum30 += cnt30[p]; sum45 += cnt45[p]; sum_cov += cov_all[p]; }
        const double rough_cvg = sum_cov / (double)num_nonzero;
        row = contig + "\t" + fmt_f64(sum_all / (double)S, 3) + "\t" + fmt_f64((double)total_bases_covered / (double)contig_len, 3) + "\t" + fmt_f64(rough_cvg, 3) + "\t" +
              std::to_string(total_bases_covered) + "\t" + fmt_f64(sum15 / (double)S, 3) + "\t" + fmt_f64(sum30 / (double)S, 3) + "\t" + fmt_f64(sum45 / (double)S, 3) + "\t" +
              fmt_f64(hq.avg_err, 4) + "\n";
    }
    // write_nosnp_reads_parts (:151-165)
    {
        std::ofstream f(dir + "/reads_without_snps.tsv", std::ios::trunc);
        f << "READ_NAME\tREAD_LENGTH_IN_BASES\n";
        for (const Frag* fr : snpless_frags) f << fr->id << "\t" << (fr->seq_len[0] + fr->seq_len[1]) << "\n";
    }
    return row;
}
void append_ploidy_row(const Options& o, const std::string& row) {
    std::ofstream pl(o.out_dir + "/contig_ploidy_info.tsv", std::ios::app);
    pl << row;
}
}  // namespace

void write_outputs(Session& s, const std::vector<std::vector<const Frag*>>& part, const std::vector<std::pair<SnpPosition, SnpPosition>>& ranges,
                   const std::string& dir, const std::string& prefix, const std::string& contig, const std::vector<GnPosition>& snp_to_gn,
                   const Options& o, const std::vector<const Frag*>& snpless_frags, size_t contig_len) {
    const size_t n = part.size();
    // ---- device: get_hapq (:40-41) and get_errors_cov_from_frags per haploset ---------------------------------------------------
    const HapqResult hq = get_hapq(s, part, snp_to_gn, ranges, o);
    std::vector<double> st(4 * n + 4, 0.0);
    {
        std::vector<uint64_t> off{0};
        std::vector<uint32_t> reads, rng;
        for (size_t g = 0; g < n; ++g) {
            for (const Frag* f : part[g]) reads.push_back((uint32_t)f->counter_id);
            off.push_back(reads.size());
            rng.push_back(ranges[g].first); rng.push_back(ranges[g].second);
        }
        const floria_hip_contig* one[1] = {s.contig()};
        if (n) check(floria_hip_haploset_stats(s.ctx(), one, 1, nullptr, off.data(), reads.data(), rng.data(), (uint32_t)n, st.data()));
    }
    append_ploidy_row(o, write_contig_files(part, ranges, dir, prefix, contig, snp_to_gn, snpless_frags, contig_len, hq, st));
    if (o.output_reads) { write_reads(part, ranges, dir, !o.trim_reads, hq.hapqs, o.gzip); write_nosnp_reads(dir, snpless_frags, o.gzip); }
}

// ---- --output-reads (file_writer.rs:86-150, 168-217, 370-560) ----------------------------------------------------------------------
namespace {
// bio::io::fastq::Writer::write(id, None, seq, qual) -> "@id\nseq\n+\nqual\n", into a plain or gzip file
struct FastqOut {
    std::string path; bool gz; FILE* f = nullptr; gzFile g = nullptr;
    FastqOut(const std::string& p, bool gzip) : path(p), gz(gzip) {
        if (gz) g = gzopen(p.c_str(), "wb"); else f = fopen(p.c_str(), "wb");
        if (!g && !f) throw Error(FLORIA_E_INVALID, "Can't create file " + p);
    }
    void put(const void* d, size_t n) { if (gz) gzwrite(g, d, (unsigned)n); else fwrite(d, 1, n, f); }
    void write(const std::string& id, const char* seq, size_t ns, const uint8_t* qual, size_t nq) {
        put("@", 1); put(id.data(), id.size()); put("\n", 1); put(seq, ns); put("\n+\n", 3); put(qual, nq); put("\n", 1);
    }
    void close() { if (g) { gzclose(g); g = nullptr; } if (f) { fclose(f); f = nullptr; } }
    ~FastqOut() { close(); }
};
const uint8_t BANG = 33;
std::string revcomp(const std::string& s) {
    std::string r(s.rbegin(), s.rend());
    for (char& c : r) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';        // (seq_string holds ACGT only)
    return r;
}
// write_paired_reads_no_trim (:168-217): mate 1 as stored, mate 2 reverse-complemented with its qualities as stored
void write_paired_reads_no_trim(FastqOut& w1, FastqOut& w2, const Frag& frag) {
    if (frag.seq_string[0].empty()) w1.write(frag.id + "/1", "N", 1, &BANG, 1);
    else w1.write(frag.id + "/1", frag.seq_string[0].data(), frag.seq_string[0].size(), frag.qual_string[0].data(), frag.qual_string[0].size());
    if (frag.seq_string[1].empty()) w2.write(frag.id + "/2", "N", 1, &BANG, 1);
    else { const std::string rc = revcomp(frag.seq_string[1]); w2.write(frag.id + "/2", rc.data(), rc.size(), frag.qual_string[1].data(), frag.qual_string[1].size()); }
}
}  // namespace

void write_nosnp_reads(const std::string& dir, const std::vector<const Frag*>& snpless_frags, bool gzip) {
    const std::string gz = gzip ? ".gz" : "";
    const std::string p0 = dir + "/long_reads/snpless.fastq" + gz, p1 = dir + "/short_reads/snpless_paired1.fastq" + gz, p2 = dir + "/short_reads/snpless_paired2.fastq" + gz;
    bool paired_written = false, single_end_written = false;
    {
        FastqOut w(p0, gzip), w1(p1, gzip), w2(p2, gzip);
        for (const Frag* frag : snpless_frags) {
            if (frag->is_paired) { paired_written = true; write_paired_reads_no_trim(w1, w2, *frag); }
            else {
                single_end_written = true;
                if (frag->seq_string[0].empty()) w.write(frag->id, "N", 1, &BANG, 1);
                else w.write(frag->id, frag->seq_string[0].data(), frag->seq_string[0].size(), frag->qual_string[0].data(), frag->qual_string[0].size());
            }
        }
    }
    if (!paired_written) { remove(p1.c_str()); remove(p2.c_str()); }
    if (!single_end_written) remove(p0.c_str());
}

void write_reads(const std::vector<std::vector<const Frag*>>& part, const std::vector<std::pair<SnpPosition, SnpPosition>>& ranges, const std::string& dir,
                 bool extend_read_clipping, const std::vector<uint8_t>& hapqs, bool gzip) {
    constexpr size_t EXTENSION_BASES = 25;                                       // constants.rs:22
    mkdir_p(dir + "/short_reads"); mkdir_p(dir + "/long_reads");
    for (size_t i = 0; i < part.size(); ++i) {
        if (part[i].empty() || ranges.empty()) continue;                         // (hapqs[i] < HAPQ_CUTOFF = 0 never holds)
        (void)hapqs;
        const SnpPosition left_snp_pos = ranges[i].first, right_snp_pos = ranges[i].second;
        std::vector<const Frag*> vec_part(part[i]);
        std::sort(vec_part.begin(), vec_part.end(), [](const Frag* a, const Frag* b) { return *a < *b; });
        const std::string gz = gzip ? ".gz" : "";
        const std::string p0 = dir + "/long_reads/" + std::to_string(i) + "_part.fastq" + gz, p1 = dir + "/short_reads/" + std::to_string(i) + "_part_paired1.fastq" + gz,
                          p2 = dir + "/short_reads/" + std::to_string(i) + "_part_paired2.fastq" + gz;
        bool paired_written = false, single_end_written = false;
        {
            FastqOut w(p0, gzip), w1(p1, gzip), w2(p2, gzip);
            for (const Frag* fp : vec_part) {
                const Frag& frag = *fp;
                if (frag.seq_string[0].empty() && frag.seq_string[1].empty()) continue;            // no primary sequence
                if (frag.first_position > right_snp_pos || frag.last_position < left_snp_pos) continue;   // fell off a merged haplogroup's range
                size_t left_seq_pos = 0;
                if (!(frag.first_position > left_snp_pos && extend_read_clipping)) {
                    auto it = frag.snp_pos_to_seq_pos.lower_bound(left_snp_pos);                  // first SNP of the read at or after the left end
                    if (it == frag.snp_pos_to_seq_pos.end()) throw Error(FLORIA_E_INVALID, "left snp position of partition for the read was not found.");
                    left_seq_pos = it->second.second;
                }
                left_seq_pos = left_seq_pos > EXTENSION_BASES ? left_seq_pos - EXTENSION_BASES : 0;
                size_t right_seq_pos; uint8_t right_read_pair;
                if (frag.last_position < right_snp_pos && extend_read_clipping) {
                    right_read_pair = frag.is_paired ? 1 : 0;
                    right_seq_pos = frag.seq_string[right_read_pair].empty() ? 0 : frag.seq_string[right_read_pair].size() - 1;
                } else {
                    auto it = frag.snp_pos_to_seq_pos.upper_bound(right_snp_pos);                 // last SNP of the read at or before the right end
                    if (it == frag.snp_pos_to_seq_pos.begin()) throw Error(FLORIA_E_INVALID, "right snp position of partition for the read was not found.");
                    --it;
                    right_seq_pos = it->second.second; right_read_pair = it->second.first;
                }
                const size_t rlen = frag.seq_string[right_read_pair].size();
                if (rlen == 0) right_seq_pos = 0;
                else if (rlen > EXTENSION_BASES + 1 && right_seq_pos < rlen - EXTENSION_BASES - 1) right_seq_pos += EXTENSION_BASES;
                else right_seq_pos = rlen - 1;
                if (frag.is_paired) { paired_written = true; write_paired_reads_no_trim(w1, w2, frag); }
                else {
                    single_end_written = true;
                    if (left_seq_pos > right_seq_pos) continue;                                  // (a read id that is not unique, or supplementary pieces)
                    const size_t n0 = frag.seq_string[0].size();
                    if (right_seq_pos + 1 > n0) throw Error(FLORIA_E_INVALID, "read " + frag.id + ": sequence position past the end of the primary alignment's sequence (the reference panics here)");
                    w.write(frag.id, frag.seq_string[0].data() + left_seq_pos, right_seq_pos + 1 - left_seq_pos, frag.qual_string[0].data() + left_seq_pos, right_seq_pos + 1 - left_seq_pos);
                }
            }
        }
        if (!paired_written) { remove(p1.c_str()); remove(p2.c_str()); }
        if (!single_end_written) remove(p0.c_str());
    }
}

// floria.rs:271-281: with --overwrite an existing contig directory is removed (remove_dir_all) before anything of the contig is written, so that
// nothing of an earlier run survives in it (read directories of --output-reads, debug dumps); without --overwrite the run has already refused an
// existing output directory
void prepare_contig_dir(const std::string& dir, const Options& o) {
    struct stat st;
    if (o.overwrite && stat(dir.c_str(), &st) == 0 && S_ISDIR(st.st_mode))
        nftw(dir.c_str(), [](const char* p, const struct stat*, int, struct FTW*) -> int { return remove(p); }, 32, FTW_DEPTH | FTW_PHYS);
}
std::string write_contig_files(const ContigWork& w, const Options& o) {
    std::string row = write_contig_files(w.final_parts, w.final_ranges, w.out_dir, w.name, w.name, *w.snp_to_genome_pos, w.snpless, w.contig_len, w.hq, w.stats);
    if (o.output_reads) {                                                                   // file_writer.rs:68-84
        write_reads(w.final_parts, w.final_ranges, w.out_dir, !o.trim_reads, w.hq.hapqs, o.gzip);
        write_nosnp_reads(w.out_dir, w.snpless, o.gzip);
    }
    return row;
}
void write_outputs(const ContigWork& w, const Options& o) { append_ploidy_row(o, write_contig_files(w, o)); }
void append_contig_ploidy_row(const Options& o, const std::string& row) { append_ploidy_row(o, row); }

}  // namespace floria

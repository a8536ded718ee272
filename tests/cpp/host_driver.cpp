// host_driver.cpp — exercises the C++ host mirror (floria_amd/host) the way floria.rs:229-366 drives the reference:
// read a pileup fixture, build `Frag`s, sort + number them, generate_hap_graph, then process_reads_for_final_parts on the
// node read sets, and print everything in a line format the pytest wrapper compares with the oracle.
//   usage: host_driver <fixture> <epsilon> <block_length> <max_ploidy> <beam>
#include <algorithm>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "../../floria_amd/host/floria_host.hpp"

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage\n"); return 2; }
    try {
        std::ifstream in(argv[1]);
        size_t n_reads, n_snps;
        in >> n_reads >> n_snps;
        std::vector<floria::GnPosition> snp_to_genome_pos(n_snps);
        for (auto& g : snp_to_genome_pos) in >> g;
        std::vector<floria::Frag> all_frags(n_reads);
        for (size_t r = 0; r < n_reads; ++r) {
            size_t L; in >> L;
            all_frags[r].id = "read" + std::to_string(r);
            all_frags[r].counter_id = r;                                  // pre-sort id (ties in Frag::cmp)
            for (size_t c = 0; c < L; ++c) { uint32_t s, a, q; in >> s >> a >> q; all_frags[r].update(s, (uint8_t)a, (uint8_t)q); }
        }
        std::sort(all_frags.begin(), all_frags.end());                    // floria.rs:289
        for (size_t i = 0; i < all_frags.size(); ++i) all_frags[i].counter_id = i;     // :290-293
        floria::Options opt;
        opt.epsilon = atof(argv[2]); opt.block_length = (size_t)atoll(argv[3]); opt.max_ploidy = (size_t)atoll(argv[4]); opt.max_number_solns = (size_t)atoll(argv[5]);
        floria::Session session(opt.device);
        auto graph = floria::generate_hap_graph(session, all_frags, snp_to_genome_pos, "", opt);
        std::vector<std::vector<const floria::Frag*>> parts;
        std::vector<std::pair<floria::SnpPosition, floria::SnpPosition>> ranges;
        for (auto& col : graph)
            for (auto& node : col) {
                printf("NODE %zu %zu %zu %u %u %.17g reads", node.column, node.row, node.id, node.snp_endpoints.first, node.snp_endpoints.second, node.cov);
                for (auto* f : node.frag_set) printf(" %zu", f->counter_id);
                printf(" out");
                for (auto& e : node.out_edges) printf(" %zu:%g", e.first, e.second);
                printf(" in");
                for (auto& e : node.in_edges) printf(" %zu:%g", e.first, e.second);
                printf("\n");
                if (!node.frag_set.empty()) { parts.push_back(node.frag_set); ranges.push_back(node.snp_endpoints); }
            }
        auto fin = floria::process_reads_for_final_parts(session, parts, {}, ranges, opt, snp_to_genome_pos);
        for (size_t g = 0; g < fin.first.size(); ++g) {
            printf("GROUP %u %u reads", fin.second[g].first, fin.second[g].second);
            for (auto* f : fin.first[g]) printf(" %zu", f->counter_id);
            printf("\n");
        }
        auto hq = floria::get_hapq(session, fin.first, snp_to_genome_pos, fin.second, opt);      // file_writer.rs:40-41
        printf("HAPQ");
        for (size_t g = 0; g < hq.hapqs.size(); ++g) printf(" %u", (unsigned)hq.hapqs[g]);
        printf("\n");
        // error behaviour: non-increasing VCF positions are fatal in the reference (utils_frags.rs:422-425)
        try {
            floria::get_range_with_lengths({10, 20, 15, 40}, 100, 33, 0.0005);
            printf("ERRCHECK missing\n");
        } catch (const floria::Error& e) { printf("ERRCHECK %d %s\n", e.code, e.what()); }
    } catch (const std::exception& e) { fprintf(stderr, "host_driver: %s\n", e.what()); return 1; }
    return 0;
}

// FlatMap (floria_amd/host/floria_host.hpp) against std::map under random operations: the subset of std::map's interface the host code uses.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>

#include "../../floria_amd/host/floria_host.hpp"

int main() {
    std::mt19937 rng(12345);
    for (int round = 0; round < 200; ++round) {
        floria::FlatMap<uint32_t, int> f;
        std::map<uint32_t, int> m;
        const int n_ops = 1 + (int)(rng() % 400);
        for (int op = 0; op < n_ops; ++op) {
            const uint32_t k = rng() % 64;
            switch (rng() % 6) {
                case 0: case 1: { const int v = (int)(rng() % 1000); f[k] = v; m[k] = v; break; }
                case 2: { if (f.erase(k) != m.erase(k)) { puts("erase(key) count differs"); return 1; } break; }
                case 3: { auto a = f.lower_bound(k); auto b = m.lower_bound(k);
                          if ((a == f.end()) != (b == m.end()) || (a != f.end() && (a->first != b->first || a->second != b->second))) { puts("lower_bound differs"); return 1; } break; }
                case 4: { auto a = f.upper_bound(k); auto b = m.upper_bound(k);
                          if ((a == f.end()) != (b == m.end()) || (a != f.end() && a->first != b->first)) { puts("upper_bound differs"); return 1; } break; }
                default: { if (f.count(k) != m.count(k)) { puts("count differs"); return 1; }
                           if (m.count(k) && f.at(k) != m.at(k)) { puts("at differs"); return 1; }
                           bool threw = false; try { (void)f.at(k); } catch (const std::out_of_range&) { threw = true; }
                           if (threw != (m.count(k) == 0)) { puts("at() exception differs"); return 1; } break; }
            }
            if (f.size() != m.size() || f.empty() != m.empty()) { puts("size differs"); return 1; }
        }
        auto b = m.begin();
        for (auto a = f.begin(); a != f.end(); ++a, ++b) if (a->first != b->first || a->second != b->second) { puts("iteration differs"); return 1; }
        if (!m.empty() && (f.rbegin()->first != m.rbegin()->first)) { puts("rbegin differs"); return 1; }
        for (auto it = f.begin(); it != f.end();) { if (it->first % 3 == 0) { m.erase(it->first); it = f.erase(it); } else ++it; }      // erase(iterator) returns the next
        if (f.size() != m.size()) { puts("erase(iterator) differs"); return 1; }
    }
    puts("OK");
    return 0;
}

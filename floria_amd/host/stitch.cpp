// stitch.cpp — global stitching of the hap graph: solve_lp_graph (solve_flow.rs:195-290) and get_disjoint_paths_rewrite
// (graph_processing.rs:462-750).  Host C++ (small sequential graph work on the CPU in the reference too); see floria_host.hpp.
//
// solve_lp_graph.  The reference hands   min sum_e t_e   s.t.  t_e >= |x_e - a_e|,  x_e >= 0,  inflow(v) == outflow(v) for every
// node v of an interior column that has both in- and out-edges   to minilp (a third-party simplex).  The constraint matrix is a
// network matrix and every a_e is an integer (a count of reads), so the LP is a min-cost flow with unit costs and its vertices
// are integral; it is solved here exactly, as a flow problem:
//     x_e = a_e + y+_e - y-_e,  y+ >= 0 (arc u->v, cost 1),  0 <= y-_e <= a_e (arc v->u, cost 1)
//     a constrained node v must emit d_v = in_a(v) - out_a(v) units of y-flow (absorb -d_v if negative)
//     every unconstrained node trades freely with a hub (cost 0), which takes up the difference of the totals
// by successive shortest augmenting paths.  The optimum VALUE equals minilp's; where the optimum is not unique the reference's
// flows are whatever vertex its simplex stops at, which cannot be reproduced without the crate (parity unpinned for such cases,
// DESIGN.md).  Non-unique optima are the norm, not the exception (a node with inflow 5 and outflow 3 can lower the one or raise the
// other at the same cost), so the solver (a) can return either extreme of its tie-breaking (LpTie) and (b) counts the edges whose
// flow differs in some other optimal solution (LpInfo::movable_edges), so that the dependence of the output on the choice is
// measurable: floria-hip --lp-tie last, scripts/lp_tie_rate.py, DESIGN.md "Stitching".
#include "floria_host.hpp"

#include <algorithm>
#include <cmath>
#include <deque>
#include <limits>

namespace floria {

namespace {

constexpr double MIN_SHARED_READS_UNAMBIG = 2.;     // constants.rs:4

struct McfArc { int to; int64_t cap; int cost; };

struct Mcf {
    std::vector<McfArc> arcs;                 // arc i and its reverse i ^ 1
    std::vector<std::vector<int>> adj;
    explicit Mcf(int n) : adj(n) {}
    int add(int u, int v, int64_t cap, int cost) {
        arcs.push_back({v, cap, cost}); adj[u].push_back((int)arcs.size() - 1);
        arcs.push_back({u, 0, -cost}); adj[v].push_back((int)arcs.size() - 1);
        return (int)arcs.size() - 2;
    }
    // potentials pi with cost + pi[u] - pi[v] >= 0 on every residual arc (Bellman-Ford from a virtual root; an optimal flow has
    // no negative residual cycle, so this converges)
    std::vector<int64_t> potentials() const {
        const int n = (int)adj.size();
        std::vector<int64_t> pi(n, 0);
        std::vector<char> inq(n, 1);
        std::deque<int> q;
        for (int v = 0; v < n; ++v) q.push_back(v);
        while (!q.empty()) {
            const int u = q.front(); q.pop_front(); inq[u] = 0;
            for (int ai : adj[u]) {
                const McfArc& a = arcs[ai];
                if (a.cap > 0 && pi[u] + a.cost < pi[a.to]) { pi[a.to] = pi[u] + a.cost; if (!inq[a.to]) { inq[a.to] = 1; q.push_back(a.to); } }
            }
        }
        return pi;
    }
    // is there a path to -> ... -> from over residual arcs of zero reduced cost that does not use `banned`?  Together with an arc
    // from -> to of zero reduced cost that closes a zero-cost residual cycle: pushing one unit around it is another optimum.
    bool zero_path(int from, int to, int banned, const std::vector<int64_t>& pi) const {
        std::vector<char> seen(adj.size(), 0);
        std::vector<int> stack{to};
        seen[to] = 1;
        while (!stack.empty()) {
            const int u = stack.back(); stack.pop_back();
            if (u == from) return true;
            for (int ai : adj[u]) {
                const McfArc& a = arcs[ai];
                if (ai == banned || a.cap <= 0 || seen[a.to] || a.cost + pi[u] - pi[a.to] != 0) continue;
                seen[a.to] = 1; stack.push_back(a.to);
            }
        }
        return false;
    }
    // successive shortest paths (SPFA: residual arcs carry negative costs); ties between equally short paths are broken by
    // the order of the adjacency lists (arc insertion order, or its reverse for LpTie::Last), so the result is deterministic
    void run(int S, int T) {
        const int n = (int)adj.size();
        const int64_t INF = std::numeric_limits<int64_t>::max() / 4;
        std::vector<int64_t> dist(n);
        std::vector<int> prev_arc(n);
        std::vector<char> inq(n);
        for (;;) {
            std::fill(dist.begin(), dist.end(), INF);
            std::fill(prev_arc.begin(), prev_arc.end(), -1);
            std::fill(inq.begin(), inq.end(), 0);
            std::deque<int> q;
            dist[S] = 0; q.push_back(S); inq[S] = 1;
            while (!q.empty()) {
                const int u = q.front(); q.pop_front(); inq[u] = 0;
                for (int ai : adj[u]) {
                    const McfArc& a = arcs[ai];
                    if (a.cap > 0 && dist[u] + a.cost < dist[a.to]) {
                        dist[a.to] = dist[u] + a.cost; prev_arc[a.to] = ai;
                        if (!inq[a.to]) { inq[a.to] = 1; q.push_back(a.to); }
                    }
                }
            }
            if (dist[T] >= INF) return;
            int64_t push = INF;
            for (int v = T; v != S; v = arcs[prev_arc[v] ^ 1].to) push = std::min(push, arcs[prev_arc[v]].cap);
            for (int v = T; v != S; v = arcs[prev_arc[v] ^ 1].to) { arcs[prev_arc[v]].cap -= push; arcs[prev_arc[v] ^ 1].cap += push; }
        }
    }
};

// ---- petgraph::stable_graph::StableGraph<(usize, usize), f64> as get_disjoint_paths_rewrite uses it ------------------------
// Directed; node and edge indices are stable under removal; a node's edge lists are singly linked with the most recently added
// edge at the head (petgraph's Graph::add_edge), removal unlinks without reordering.  Iteration orders below are the crate's
// (0.6.5, Cargo.lock) as published; they cannot be verified here (no Rust toolchain) and only break exact ties.
struct StableGraph {
    static constexpr int END = -1;
    struct Node { bool alive; size_t col, row; int head[2]; };         // head[0] outgoing, head[1] incoming
    struct Edge { bool alive; int node[2]; double w; int next[2]; };   // node[0] source, node[1] target
    std::vector<Node> nodes;
    std::vector<Edge> edges;
    size_t n_alive = 0;
    int add_node(size_t col, size_t row) { nodes.push_back({true, col, row, {END, END}}); ++n_alive; return (int)nodes.size() - 1; }
    int add_edge(int a, int b, double w) {
        Edge e{true, {a, b}, w, {nodes[a].head[0], nodes[b].head[1]}};
        edges.push_back(e);
        const int ei = (int)edges.size() - 1;
        nodes[a].head[0] = ei; nodes[b].head[1] = ei;
        return ei;
    }
    void unlink(int ei, int k) {
        const int n = edges[ei].node[k];
        int* p = &nodes[n].head[k];
        while (*p != END && *p != ei) p = &edges[*p].next[k];
        if (*p == ei) *p = edges[ei].next[k];
    }
    void remove_edge(int ei) {
        if (ei < 0 || ei >= (int)edges.size() || !edges[ei].alive) return;
        unlink(ei, 0); unlink(ei, 1);
        edges[ei].alive = false;
    }
    void remove_node(int n) {
        if (!nodes[n].alive) return;
        for (int k = 0; k < 2; ++k) while (nodes[n].head[k] != END) remove_edge(nodes[n].head[k]);
        nodes[n].alive = false; --n_alive;
    }
    size_t in_degree(int n) const { size_t c = 0; for (int e = nodes[n].head[1]; e != END; e = edges[e].next[1]) ++c; return c; }
    size_t out_degree(int n) const { size_t c = 0; for (int e = nodes[n].head[0]; e != END; e = edges[e].next[0]) ++c; return c; }
    // petgraph::algo::toposort (kosaraju-style DFS over node_identifiers().rev(), neighbours pushed in list order)
    std::vector<int> toposort() const {
        std::vector<char> discovered(nodes.size(), 0), finished(nodes.size(), 0);
        std::vector<int> stack, finish;
        for (int i = (int)nodes.size() - 1; i >= 0; --i) {
            if (!nodes[i].alive || discovered[i]) continue;
            stack.push_back(i);
            while (!stack.empty()) {
                const int nx = stack.back();
                if (!discovered[nx]) {
                    discovered[nx] = 1;
                    for (int e = nodes[nx].head[0]; e != END; e = edges[e].next[0]) { const int succ = edges[e].node[1]; if (!discovered[succ]) stack.push_back(succ); }
                } else {
                    stack.pop_back();
                    if (!finished[nx]) { finished[nx] = 1; finish.push_back(nx); }
                }
            }
        }
        std::reverse(finish.begin(), finish.end());
        return finish;
    }
};

struct TraceBackNode { double score = 0.; int prev_ind = -1; bool is_sink = false, is_source = false; };

}  // namespace

// solve_flow.rs:195-290
FlowUpVec solve_lp_graph(const std::vector<std::vector<HapNode>>& hap_graph, LpTie tie, LpInfo* info) {
    // edges in the reference's order: nodes by (column, row), each node's out_edges in order (:211-226)
    struct E { size_t c1, r1, c2, r2; int64_t a; };
    std::vector<E> edges;
    std::vector<std::vector<size_t>> node_id(hap_graph.size());
    size_t n_nodes = 0;
    for (size_t c = 0; c < hap_graph.size(); ++c) { node_id[c].resize(hap_graph[c].size()); for (size_t r = 0; r < hap_graph[c].size(); ++r) node_id[c][r] = n_nodes++; }
    for (size_t c = 0; c < hap_graph.size(); ++c)
        for (size_t r = 0; r < hap_graph[c].size(); ++r)
            for (const auto& oe : hap_graph[c][r].out_edges) edges.push_back({c, r, c + 1, oe.first, (int64_t)std::llround(oe.second)});
    // flow conservation only at nodes of interior columns with both in- and out-edges (:235-239)
    std::vector<char> constrained(n_nodes, 0);
    for (size_t c = 1; c + 1 < hap_graph.size(); ++c)
        for (size_t r = 0; r < hap_graph[c].size(); ++r)
            if (!hap_graph[c][r].in_edges.empty() && !hap_graph[c][r].out_edges.empty()) constrained[node_id[c][r]] = 1;
    std::vector<int64_t> d(n_nodes, 0);
    for (const E& e : edges) { d[node_id[e.c2][e.r2]] += e.a; d[node_id[e.c1][e.r1]] -= e.a; }
    const int HUB = (int)n_nodes, S = HUB + 1, T = HUB + 2;
    Mcf g((int)n_nodes + 3);
    const int64_t BIG = std::numeric_limits<int64_t>::max() / 8;
    std::vector<int> arc_plus(edges.size()), arc_minus(edges.size());
    for (size_t i = 0; i < edges.size(); ++i) {
        const int u = (int)node_id[edges[i].c1][edges[i].r1], v = (int)node_id[edges[i].c2][edges[i].r2];
        arc_plus[i] = g.add(u, v, BIG, 1);
        arc_minus[i] = g.add(v, u, edges[i].a, 1);
    }
    int64_t sup = 0, dem = 0;
    for (size_t v = 0; v < n_nodes; ++v) {
        if (!constrained[v]) { g.add((int)v, HUB, BIG, 0); g.add(HUB, (int)v, BIG, 0); continue; }
        if (d[v] > 0) { g.add(S, (int)v, d[v], 0); sup += d[v]; }
        else if (d[v] < 0) { g.add((int)v, T, -d[v], 0); dem += -d[v]; }
    }
    if (sup > dem) g.add(HUB, T, sup - dem, 0);
    else if (dem > sup) g.add(S, HUB, dem - sup, 0);
    if (tie == LpTie::Last) for (auto& l : g.adj) std::reverse(l.begin(), l.end());
    g.run(S, T);
    FlowUpVec out;
    out.reserve(edges.size());
    int64_t cost = 0;
    for (size_t i = 0; i < edges.size(); ++i) {
        const int64_t yp = g.arcs[arc_plus[i] ^ 1].cap, ym = g.arcs[arc_minus[i] ^ 1].cap;      // flow on an arc = capacity of its reverse
        out.push_back({{edges[i].c1, edges[i].r1}, {edges[i].c2, edges[i].r2}, (double)(edges[i].a + yp - ym)});
        cost += yp + ym;
    }
    if (info) {
        // x_e is not determined by optimality iff one of the four residual arcs of e (y+, y-, and their reverses) lies on a zero-cost
        // residual cycle other than the trivial one with its own reverse.  (Two different arcs between the same pair of nodes cannot
        // close a zero-cost 2-cycle: y+ with y- costs 2, and the two reverses are never both residual at an optimum.)
        const std::vector<int64_t> pi = g.potentials();
        info->cost = cost;
        info->movable_edges = 0;
        for (size_t i = 0; i < edges.size(); ++i) {
            bool movable = false;
            for (int ai : {arc_plus[i], arc_plus[i] ^ 1, arc_minus[i], arc_minus[i] ^ 1}) {
                const McfArc& a = g.arcs[ai];
                const int from = g.arcs[ai ^ 1].to;
                if (a.cap <= 0 || a.cost + pi[from] - pi[a.to] != 0) continue;
                if (g.zero_path(from, a.to, ai ^ 1, pi)) { movable = true; break; }
            }
            info->movable_edges += movable;
        }
    }
    return out;
}

// graph_processing.rs:462-750 (do_binning, a hidden flag, is not supported)
std::pair<std::vector<std::vector<const Frag*>>, std::vector<std::pair<SnpPosition, SnpPosition>>> get_disjoint_paths_rewrite(
    std::vector<std::vector<HapNode>>& hap_graph, const FlowUpVec& flow_update_vec, const Options&) {
    StableGraph pg;
    for (auto& col : hap_graph) for (auto& n : col) n.out_flows.clear();
    for (const auto& fu : flow_update_vec) {                                        // :474-483
        if (fu.flow < MIN_SHARED_READS_UNAMBIG) continue;
        hap_graph[fu.n1.first][fu.n1.second].out_flows.push_back({fu.n2.second, fu.flow});
    }
    std::vector<std::vector<int>> index(hap_graph.size());
    for (size_t c = 0; c < hap_graph.size(); ++c) for (size_t r = 0; r < hap_graph[c].size(); ++r) index[c].push_back(pg.add_node(c, r));   // :485-492
    for (size_t c = 0; c < hap_graph.size(); ++c)                                   // :495-503
        for (size_t r = 0; r < hap_graph[c].size(); ++r)
            for (const auto& of : hap_graph[c][r].out_flows) pg.add_edge(index[c][r], index[c + 1][of.first], of.second);
    const size_t num_starting_nodes = pg.nodes.size();
    std::vector<std::vector<const Frag*>> all_joined_path_parts;
    std::vector<std::pair<SnpPosition, SnpPosition>> path_parts_snp_endpoints;
    const double F64_MAX = std::numeric_limits<double>::max();
    while (pg.n_alive > 0) {
        std::vector<TraceBackNode> tb(num_starting_nodes);                           // :505-536 / :557-589
        for (size_t i = 0; i < pg.nodes.size(); ++i) if (pg.nodes[i].alive) {
            const bool is_source = pg.in_degree((int)i) == 0, is_sink = pg.out_degree((int)i) == 0;
            tb[i].score = is_source ? F64_MAX : 0.; tb[i].prev_ind = -1; tb[i].is_sink = is_sink; tb[i].is_source = is_source;
        }
        // topological sweep: widest (max-min flow) path, cutting a thin side branch off a thick path (:591-646)
        const std::vector<int> top_order = pg.toposort();
        std::vector<int> flow_cut_edges;
        for (int node : top_order)
            for (int e = pg.nodes[node].head[0]; e != StableGraph::END; e = pg.edges[e].next[0]) {
                const int source = pg.edges[e].node[0], target = pg.edges[e].node[1];
                const double flow = pg.edges[e].w;
                if (std::min(tb[source].score, flow) > tb[target].score) {
                    if (flow < tb[source].score * 0.33 && !tb[source].is_source) {
                        if (pg.in_degree(source) == 1) flow_cut_edges.push_back(e);
                        if (pg.in_degree(target) == 1) { tb[target].score = F64_MAX; tb[target].is_source = true; }
                    } else {
                        tb[target].score = std::min(tb[source].score, flow);
                        tb[target].prev_ind = source;
                    }
                }
            }
        for (int e : flow_cut_edges) pg.remove_edge(e);                              // :648-650
        int best_end = -1;                                                           // :652-667
        double best_score = std::numeric_limits<double>::lowest();                   // f64::MIN
        for (size_t i = 0; i < tb.size(); ++i) if (tb[i].score > best_score && tb[i].is_sink) { best_end = (int)i; best_score = tb[i].score; }
        if (best_end < 0) throw Error(FLORIA_E_INVALID, "get_disjoint_paths_rewrite: no sink (the reference panics here: \"Shouldn't get here\")");
        std::vector<const Frag*> joined;
        std::pair<SnpPosition, SnpPosition> ends{std::numeric_limits<SnpPosition>::max(), std::numeric_limits<SnpPosition>::min()};
        std::vector<int> best_path;
        for (int cur = best_end; cur >= 0; cur = tb[cur].prev_ind) {                 // :676-705
            const HapNode& hn = hap_graph[pg.nodes[cur].col][pg.nodes[cur].row];
            ends.first = std::min(ends.first, hn.snp_endpoints.first);
            ends.second = std::max(ends.second, hn.snp_endpoints.second);
            joined.insert(joined.end(), hn.frag_set.begin(), hn.frag_set.end());
            best_path.push_back(cur);
        }
        for (int n : best_path) pg.remove_node(n);                                   // :714-717
        std::sort(joined.begin(), joined.end(), [](const Frag* a, const Frag* b) { return a->counter_id < b->counter_id; });   // a set: each read once
        joined.erase(std::unique(joined.begin(), joined.end()), joined.end());
        all_joined_path_parts.push_back(std::move(joined));
        path_parts_snp_endpoints.push_back(ends);
    }
    return {std::move(all_joined_path_parts), std::move(path_parts_snp_endpoints)};
}

// part_block_manip.rs:622-675.  rust-lapper's count(start, stop) counts intervals [s, e) with s < stop && e > start.
std::vector<const Frag*> get_frags_in_snpless_gaps(const std::vector<std::pair<SnpPosition, SnpPosition>>& path_parts, const std::vector<GnPosition>& snp_to_gn_pos,
                                                   const std::vector<Frag>& snpless_frags, GnPosition block_len, const std::vector<Frag>& final_frags) {
    bool paired = false;
    for (const Frag& f : snpless_frags) { if (f.is_paired) paired = true; else if (paired) break; }
    std::vector<std::pair<GnPosition, GnPosition>> iv;
    for (const auto& range : path_parts) {
        GnPosition start = snp_to_gn_pos[range.first - 1];
        if (start > block_len && paired) start -= block_len;
        const GnPosition end = snp_to_gn_pos[range.second - 1] + 1 + (paired ? block_len : 0);
        iv.push_back({start, end});
    }
    std::sort(iv.begin(), iv.end());
    std::vector<GnPosition> max_end(iv.size());                    // prefix maximum of the ends: count == 0 <=> no interval overlaps
    for (size_t i = 0; i < iv.size(); ++i) max_end[i] = std::max(iv[i].second, i ? max_end[i - 1] : 0);
    auto no_overlap = [&](GnPosition start, GnPosition stop) {
        // intervals with s < stop are a prefix of the sorted list; one of them overlaps iff its end > start
        const size_t k = std::lower_bound(iv.begin(), iv.end(), std::make_pair(stop, (GnPosition)0)) - iv.begin();
        return k == 0 || !(max_end[k - 1] > start);
    };
    std::vector<const Frag*> out;
    for (const Frag& f : snpless_frags) if (no_overlap(f.first_pos_base, f.last_pos_base)) out.push_back(&f);
    for (const Frag& f : final_frags) if (no_overlap(f.first_pos_base, f.last_pos_base)) out.push_back(&f);
    return out;
}

}  // namespace floria

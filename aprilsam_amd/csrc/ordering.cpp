// ordering.cpp — nested-dissection ordering of the pose graph (host, C++).
//
// Replaces the reference's heap_minimum_degree_ordering (aprilsam.c:999-1249).  That routine need not
// be reproduced: solver results are ordering independent to ~1e-10.  What the GPU
// needs instead is a SHALLOW, BUSHY elimination tree (few dependent levels => few dependent kernel
// launches) whose nodes are dense blocks: exactly what nested dissection delivers.
//
// Algorithm per connected region: two candidate bisections — (A) median split along the principal axis
// of the pose positions (pose graphs are spatial: loop closures join nearby poses), (B) a BFS level
// structure from a pseudo-peripheral vertex — each turned from an edge cut into a MINIMUM vertex
// separator through Koenig's theorem (Hopcroft-Karp matching on the cut's bipartite graph); the better
// one wins.  Regions of <= leaf_nodes poses become dense leaves.
#include "plan.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>

namespace asam {
namespace {

struct HopcroftKarp {
    int nl = 0, nr = 0;
    std::vector<int> ptr, idx;        // left -> right adjacency
    std::vector<int> ml, mr, dist, it, q;
    std::vector<char> zl, zr;

    bool bfs() {
        q.clear();
        bool found = false;
        for (int u = 0; u < nl; u++) { if (ml[u] < 0) { dist[u] = 0; q.push_back(u); } else dist[u] = -1; }
        for (size_t h = 0; h < q.size(); h++) {
            int u = q[h];
            for (int e = ptr[u]; e < ptr[u + 1]; e++) {
                int w = mr[idx[e]];
                if (w < 0) found = true;
                else if (dist[w] < 0) { dist[w] = dist[u] + 1; q.push_back(w); }
            }
        }
        return found;
    }
    bool dfs(int u) {
        for (int &e = it[u]; e < ptr[u + 1]; e++) {
            int v = idx[e], w = mr[v];
            if (w < 0 || (dist[w] == dist[u] + 1 && dfs(w))) { ml[u] = v; mr[v] = u; return true; }
        }
        dist[u] = -1;
        return false;
    }
    void run() {
        ml.assign(nl, -1); mr.assign(nr, -1); dist.assign(nl, -1); it.assign(nl, 0);
        while (bfs()) {
            for (int u = 0; u < nl; u++) it[u] = ptr[u];
            for (int u = 0; u < nl; u++) if (ml[u] < 0) dfs(u);
        }
    }
    // minimum vertex cover (Koenig): inL[u] / inR[v] = 1 if in cover
    void cover(std::vector<char> &inL, std::vector<char> &inR) {
        zl.assign(nl, 0); zr.assign(nr, 0);
        q.clear();
        for (int u = 0; u < nl; u++) if (ml[u] < 0) { zl[u] = 1; q.push_back(u); }
        for (size_t h = 0; h < q.size(); h++) {
            int u = q[h];
            for (int e = ptr[u]; e < ptr[u + 1]; e++) {
                int v = idx[e];
                if (ml[u] == v || zr[v]) continue;      // follow NON-matching edges left->right
                zr[v] = 1;
                int w = mr[v];                          // matching edge right->left
                if (w >= 0 && !zl[w]) { zl[w] = 1; q.push_back(w); }
            }
        }
        inL.assign(nl, 0); inR.assign(nr, 0);
        for (int u = 0; u < nl; u++) inL[u] = !zl[u];
        for (int v = 0; v < nr; v++) inR[v] = zr[v];
    }
};

// unit-capacity max-flow (Dinic) used for exact vertex-separator refinement inside a band.  Flat arrays
// (linked adjacency lists) that are reused across calls: the ordering runs on every batch fall-back of
// the incremental path, so its constant factors are on a hot host path.
struct Dinic {
    struct E { int to, cap, next; };
    std::vector<E> e; std::vector<int> head, lvl, it, q;
    void reset(int n) { e.clear(); head.assign(n, -1); lvl.assign(n, -1); it.assign(n, -1); q.clear(); q.reserve(n); }
    void add(int u, int v, int c) {
        e.push_back({ v, c, head[u] }); head[u] = (int)e.size() - 1;
        e.push_back({ u, 0, head[v] }); head[v] = (int)e.size() - 1;
    }
    bool bfs(int s, int t) {
        std::fill(lvl.begin(), lvl.end(), -1);
        q.clear(); q.push_back(s); lvl[s] = 0;
        for (size_t h = 0; h < q.size(); h++)
            for (int id = head[q[h]]; id >= 0; id = e[id].next)
                if (e[id].cap > 0 && lvl[e[id].to] < 0) { lvl[e[id].to] = lvl[q[h]] + 1; q.push_back(e[id].to); }
        return lvl[t] >= 0;
    }
    int dfs(int u, int t, int f) {
        if (u == t) return f;
        for (int &id = it[u]; id >= 0; id = e[id].next) {
            if (e[id].cap > 0 && lvl[e[id].to] == lvl[u] + 1) {
                int d = dfs(e[id].to, t, std::min(f, e[id].cap));
                if (d > 0) { e[id].cap -= d; e[id ^ 1].cap += d; return d; }
            }
        }
        return 0;
    }
    int run(int s, int t) {
        int flow = 0;
        while (bfs(s, t)) { it = head; while (int f = dfs(s, t, 1 << 29)) flow += f; }
        return flow;
    }
};

// per-vertex working state of one worker
struct Shared {
    std::vector<int> label, dist, loc; std::vector<char> side; int next_label = 0;
    explicit Shared(int N) : label(N, 0), dist(N, -1), loc(N, -1), side(N, 0) {}
};
struct Item { std::vector<int> verts; int parent; };

struct Dissector {
    int N;
    const std::vector<int> &ap, &ai;
    const double *xy;
    int leaf;
    NDTree &tree;               // the (sub)tree this worker appends to: private per worker
    // Per-vertex scratch of this worker (every access to dist / loc / side of a neighbour is guarded by
    // "label[neighbour] == my region"; vertices outside the worker's regions keep label 0).
    Shared &sh;
    int *dist, *loc; char *side;
    Dinic dinic;                // scratch of refine_band
    HopcroftKarp hk_;           // scratch of cut_to_separator (buffers reused across the hundreds of calls of one plan)
    std::vector<int> B0_, B1_, band_; std::vector<char> inL_, inR_; std::vector<std::pair<double, int>> pr_;

    Dissector(int N_, const std::vector<int> &ap_, const std::vector<int> &ai_, const double *xy_, int leaf_, NDTree &t, Shared &sh_)
        : N(N_), ap(ap_), ai(ai_), xy(xy_), leaf(leaf_), tree(t), sh(sh_), dist(sh_.dist.data()), loc(sh_.loc.data()), side(sh_.side.data()) {}
    int lab(int v) const { return sh.label[v]; }
    void set_lab(int v, int L) { sh.label[v] = L; }
    int new_label() { return ++sh.next_label; }

    int new_node(std::vector<int> &&verts, int parent) {
        int id = (int)tree.nodes.size();
        tree.nodes.emplace_back();
        tree.nodes.back().verts = std::move(verts);
        if (parent < 0) tree.roots.push_back(id); else tree.nodes[parent].children.push_back(id);
        return id;
    }

    // BFS inside region L from s; returns visit order; dist[] filled for visited vertices.
    void bfs(int s, int L, std::vector<int> &order) {
        order.clear(); order.push_back(s); dist[s] = 0;
        for (size_t h = 0; h < order.size(); h++) {
            int u = order[h];
            for (int e = ap[u]; e < ap[u + 1]; e++) {
                int v = ai[e];
                if (lab(v) == L && dist[v] < 0) { dist[v] = dist[u] + 1; order.push_back(v); }
            }
        }
    }

    struct Split { std::vector<int> S, P0, P1; double cost = 1e300; bool ok = false; };

    // side[v] in {0,1} given for all v of comp: edge cut -> minimum vertex separator
    void cut_to_separator(const std::vector<int> &comp, int L, Split &out) {
        std::vector<int> &B0 = B0_, &B1 = B1_; B0.clear(); B1.clear();
        for (int v : comp) {
            bool b = false;
            for (int e = ap[v]; e < ap[v + 1] && !b; e++) { int w = ai[e]; b = (lab(w) == L && side[w] != side[v]); }
            if (b) { if (side[v] == 0) { loc[v] = (int)B0.size(); B0.push_back(v); } else { loc[v] = (int)B1.size(); B1.push_back(v); } }
        }
        if (B0.empty() || B1.empty()) { out.ok = false; return; }
        HopcroftKarp &hk = hk_; hk.nl = (int)B0.size(); hk.nr = (int)B1.size();
        hk.ptr.assign(hk.nl + 1, 0); hk.idx.clear();
        for (int i = 0; i < hk.nl; i++) {
            int v = B0[i];
            for (int e = ap[v]; e < ap[v + 1]; e++) { int w = ai[e]; if (lab(w) == L && side[w] == 1) hk.idx.push_back(loc[w]); }
            hk.ptr[i + 1] = (int)hk.idx.size();
        }
        hk.run();
        std::vector<char> &inL = inL_, &inR = inR_; hk.cover(inL, inR);
        // mark separator members with loc = -2 (loc is reset to -1 below)
        for (int i = 0; i < hk.nl; i++) loc[B0[i]] = inL[i] ? -2 : -1;
        for (int i = 0; i < hk.nr; i++) loc[B1[i]] = inR[i] ? -2 : -1;
        out.S.clear(); out.P0.clear(); out.P1.clear();
        for (int v : comp) {
            if (loc[v] == -2) { out.S.push_back(v); loc[v] = -1; }
            else if (side[v] == 0) out.P0.push_back(v); else out.P1.push_back(v);
        }
        score(comp, out);
    }

    // cost of a split: separator size, with a penalty once the larger part exceeds 62 % — on the GPU the depth of the
    // elimination tree is paid in dependent kernel launches, so balance beats a few poses.  The weights were tuned against
    // the iteration time of M3500 through a per-level cost model of the kernels (tools/nd_tune.py): lowering the quadratic
    // term from 400 to 100 alone took M3500 from 0.53 to 0.43 ms per iteration (sum over levels of the widest supernode
    // 151 -> 131 poses) and left the lattices unchanged.  APRILSAM_AMD_ND_* override them for experiments.
    void score(const std::vector<int> &comp, Split &out) {
        out.ok = !out.P0.empty() && !out.P1.empty() && !out.S.empty();
        if (out.ok) {
            double n = (double)comp.size() - (double)out.S.size();
            double imb = std::max(out.P0.size(), out.P1.size()) / n;       // 0.5 .. 1
            static const double T_IMB = getenv("APRILSAM_AMD_ND_IMB") ? atof(getenv("APRILSAM_AMD_ND_IMB")) : 0.62;         // tuning knobs (defaults measured on M3500)
            static const double T_LIN = getenv("APRILSAM_AMD_ND_LIN") ? atof(getenv("APRILSAM_AMD_ND_LIN")) : 25.0;
            static const double T_QUAD = getenv("APRILSAM_AMD_ND_QUAD") ? atof(getenv("APRILSAM_AMD_ND_QUAD")) : 100.0;
            double over = std::max(0.0, imb - T_IMB);
            out.cost = (double)out.S.size() * (1.0 + T_LIN * over) + T_QUAD * over * over * (double)comp.size();
        }
    }

    // Exact refinement inside a band: vertices within `width` hops of the separator are free, the rest of
    // P0 / P1 is contracted into source / sink, and a minimum VERTEX cut of the band (node-split unit
    // capacities, Dinic) replaces the separator.  Cleans up the ragged cuts that noisy pose positions give.
    void refine_band(const std::vector<int> &comp, int L, Split &sp, int width) {
        if (!sp.ok) return;
        for (int v : sp.P0) side[v] = 0;
        for (int v : sp.P1) side[v] = 1;
        for (int v : sp.S) side[v] = 2;
        std::vector<int> &band = band_; band.clear();      // BFS from the separator, depth <= width
        for (int v : sp.S) { dist[v] = 0; band.push_back(v); }
        for (size_t h = 0; h < band.size(); h++) {
            int u = band[h];
            if (dist[u] == width) continue;
            for (int e = ap[u]; e < ap[u + 1]; e++) { int v = ai[e]; if (lab(v) == L && dist[v] < 0) { dist[v] = dist[u] + 1; band.push_back(v); } }
        }
        const int nb = (int)band.size();
        bool core0 = false, core1 = false;
        for (int i = 0; i < nb; i++) loc[band[i]] = i;
        Dinic &fl = dinic; fl.reset(2 * nb + 2);
        const int SRC = 2 * nb, SNK = 2 * nb + 1, INF = 1 << 28;
        for (int i = 0; i < nb; i++) {
            int u = band[i];
            fl.add(2 * i, 2 * i + 1, 1);
            bool a0 = false, a1 = false;
            for (int e = ap[u]; e < ap[u + 1]; e++) {
                int v = ai[e];
                if (lab(v) != L) continue;
                if (dist[v] >= 0) fl.add(2 * i + 1, 2 * loc[v], INF);          // band -> band
                else if (side[v] == 0) a0 = true; else a1 = true;              // neighbour in a contracted core
            }
            if (a0) { fl.add(SRC, 2 * i, INF); core0 = true; }
            if (a1) { fl.add(2 * i + 1, SNK, INF); core1 = true; }
        }
        Split out;
        if (core0 && core1) {
            fl.run(SRC, SNK);
            fl.bfs(SRC, SNK);                             // residual reachability in fl.lvl
            for (int v : comp) {
                if (dist[v] < 0) { (side[v] == 0 ? out.P0 : out.P1).push_back(v); continue; }
                int i = loc[v];
                bool rin = fl.lvl[2 * i] >= 0, rout = fl.lvl[2 * i + 1] >= 0;
                if (rin && !rout) out.S.push_back(v); else if (rout) out.P0.push_back(v); else out.P1.push_back(v);
            }
            score(comp, out);
        }
        for (int v : band) { dist[v] = -1; loc[v] = -1; }
        if (out.ok && out.cost < sp.cost) sp = std::move(out);
    }

    void split_bfs(const std::vector<int> &comp, int L, Split &out) {
        std::vector<int> order;
        int s = comp[0];
        for (int sweep = 0; sweep < 2; sweep++) {           // pseudo-peripheral start
            bfs(s, L, order);
            s = order.back();
            for (int v : order) dist[v] = -1;
        }
        bfs(s, L, order);
        int maxd = dist[order.back()];
        if (maxd < 2) { for (int v : order) dist[v] = -1; out.ok = false; return; }
        // order is sorted by dist: choose the level boundary nearest to half
        size_t half = comp.size() / 2;
        int c = dist[order[half]];
        // candidates c and c+1: number of vertices with dist < c
        auto count_lt = [&](int cc) { size_t lo = 0, hi = order.size(); while (lo < hi) { size_t m = (lo + hi) / 2; if (dist[order[m]] < cc) lo = m + 1; else hi = m; } return lo; };
        size_t n0 = count_lt(c), n1 = count_lt(c + 1);
        if (c < 1 || (c + 1 <= maxd && (half - n0) > (n1 - half))) c = c + 1;
        if (c < 1) c = 1;
        if (c > maxd) c = maxd;
        for (int v : comp) side[v] = dist[v] < c ? 0 : 1;
        for (int v : order) dist[v] = -1;
        cut_to_separator(comp, L, out);
    }

    void split_geometric(const std::vector<int> &comp, int L, Split &out, double rot = 0.0) {
        if (!xy) { out.ok = false; return; }
        double mx = 0, my = 0; size_t n = comp.size();
        for (int v : comp) { mx += xy[2 * v]; my += xy[2 * v + 1]; }
        mx /= n; my /= n;
        double sxx = 0, sxy = 0, syy = 0;
        for (int v : comp) { double dx = xy[2 * v] - mx, dy = xy[2 * v + 1] - my; sxx += dx * dx; sxy += dx * dy; syy += dy * dy; }
        if (!(std::isfinite(sxx) && std::isfinite(syy) && std::isfinite(sxy))) { out.ok = false; return; }
        // principal direction of the 2x2 covariance
        double th = 0.5 * std::atan2(2 * sxy, sxx - syy) + rot;
        double ux = std::cos(th), uy = std::sin(th);
        std::vector<std::pair<double, int>> &pr = pr_; pr.resize(n);
        for (size_t i = 0; i < n; i++) { int v = comp[i]; pr[i] = { (xy[2 * v] - mx) * ux + (xy[2 * v + 1] - my) * uy, v }; }
        std::nth_element(pr.begin(), pr.begin() + n / 2, pr.end());
        for (size_t i = 0; i < n; i++) side[pr[i].second] = i < n / 2 ? 0 : 1;
        cut_to_separator(comp, L, out);
    }

    // dissect the regions on `stack` to completion.  defer_below > 0: regions of at most that many vertices are not
    // processed but moved to `deferred` (in a deterministic order) for the parallel phase.
    void run(std::vector<Item> stack, size_t defer_below = 0, std::vector<Item> *deferred = nullptr) {
        std::vector<int> order;
        while (!stack.empty()) {
            Item item = std::move(stack.back()); stack.pop_back();
            if (deferred && item.verts.size() <= defer_below) { deferred->push_back(std::move(item)); continue; }
            int L = new_label();
            for (int v : item.verts) set_lab(v, L);
            // connected components of the region
            for (int s : item.verts) {
                if (lab(s) != L) continue;
                bfs(s, L, order);
                std::vector<int> comp(order);
                for (int v : comp) dist[v] = -1;
                int Lc = new_label();
                for (int v : comp) set_lab(v, Lc);
                std::sort(comp.begin(), comp.end());
                handle(comp, Lc, item.parent, stack);
            }
        }
    }

    template <class Stack>
    void handle(std::vector<int> &comp, int L, int parent, Stack &stack) {
        if ((int)comp.size() <= leaf) { for (int v : comp) set_lab(v, -1); new_node(std::move(comp), parent); return; }
        Split cand[8];
        split_geometric(comp, L, cand[0]);
        static const int T_DIRS = getenv("APRILSAM_AMD_ND_DIRS") ? atoi(getenv("APRILSAM_AMD_ND_DIRS")) : 8;
        static const int T_REF = getenv("APRILSAM_AMD_ND_REF") ? atoi(getenv("APRILSAM_AMD_ND_REF")) : 4;
        static const int T_BAND = getenv("APRILSAM_AMD_ND_BAND") ? atoi(getenv("APRILSAM_AMD_ND_BAND")) : 2;
        if ((int)comp.size() > T_DIRS * leaf) {                         // the extra directions only pay near the top of the tree
            split_geometric(comp, L, cand[1], 1.5707963267948966);      // orthogonal axis
            split_geometric(comp, L, cand[2], 0.7853981633974483);      // diagonal
        }
        static const int T_MORE = getenv("APRILSAM_AMD_ND_MORE") ? atoi(getenv("APRILSAM_AMD_ND_MORE")) : 0;      // region size (x leaf) above which 4 more directions are tried
        if (T_MORE > 0 && (int)comp.size() > T_MORE * leaf) {
            split_geometric(comp, L, cand[4], 2.356194490192345);       // other diagonal
            split_geometric(comp, L, cand[5], 0.39269908169872414);
            split_geometric(comp, L, cand[6], 1.1780972450961724);
            split_geometric(comp, L, cand[7], 1.9634954084936207);
        }
        split_bfs(comp, L, cand[3]);
        Split *best = nullptr, *second = nullptr;
        for (Split &c : cand) {
            if (!c.ok) continue;
            if (!best || c.cost < best->cost) { second = best; best = &c; }
            else if (!second || c.cost < second->cost) second = &c;
        }
        if (best && (int)comp.size() > T_REF * leaf) {
            for (Split *c : { best, second }) {
                if (!c || (c == second && second->cost > 1.25 * best->cost)) continue;
                for (int pass = 0; pass < 2; pass++) { double before = c->cost; refine_band(comp, L, *c, T_BAND); if (c->cost >= before) break; }
            }
            if (second && second->cost < best->cost) best = second;
        }
        if (!best) { for (int v : comp) set_lab(v, -1); new_node(std::move(comp), parent); return; }   // dense region
        for (int v : best->S) set_lab(v, -1);
        int t = new_node(std::move(best->S), parent);
        stack.push_back({ std::move(best->P0), t });
        stack.push_back({ std::move(best->P1), t });
    }
};

}  // namespace

void nested_dissection(int N, const std::vector<int> &adj_ptr, const std::vector<int> &adj,
                       const double *xy, int leaf_nodes, NDTree &tree) {
    tree.nodes.clear(); tree.roots.clear();
    if (N <= 0) return;
    if (leaf_nodes < 1) leaf_nodes = 1;
    Shared sh(N);
    Item all; all.verts.resize(N); std::iota(all.verts.begin(), all.verts.end(), 0); all.parent = -1;
    std::vector<Item> top; top.push_back(std::move(all));
    // Phase 1 (this thread): the top of the tree.  Regions of at most N/12 vertices are set aside, in a fixed order.
    // Phase 2: every such region is dissected into a PRIVATE subtree by a pool of threads (regions are vertex-disjoint),
    // and the subtrees are appended to the tree in that fixed order -- so the result does not depend on the number of
    // threads or on their timing (all ranks of a sharded run and every re-run build the identical plan).
    std::vector<Item> tasks;
    const size_t defer_below = N >= 1024 ? (size_t)N / 12 : 0;
    {
        Dissector d(N, adj_ptr, adj, xy, leaf_nodes, tree, sh);
        d.run(std::move(top), defer_below, defer_below ? &tasks : nullptr);
    }
    if (tasks.empty()) return;
    const int nt = (int)tasks.size();
    std::vector<NDTree> sub(nt);
    auto work = [&](int i) {
        std::vector<Item> st; Item it; it.verts = std::move(tasks[i].verts); it.parent = -1; st.push_back(std::move(it));
        Shared mine(N);           // private per-vertex state: sharing one set of arrays between threads is correct (disjoint
        Dissector d(N, adj_ptr, adj, xy, leaf_nodes, sub[i], mine);      // vertices) but falsely shares cache lines -- measured: no speed-up
        d.run(std::move(st));
    };
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = std::max(1, std::min({ nt, 16, hw ? (int)hw : 1 }));
    if (const char *e = getenv("APRILSAM_AMD_PLAN_THREADS")) nthreads = std::max(1, atoi(e));      // 1 = single-threaded planning
    if (nthreads == 1) { for (int i = 0; i < nt; i++) work(i); }
    else {
        std::atomic<int> next{ 0 };
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; t++) pool.emplace_back([&] { for (int i; (i = next.fetch_add(1)) < nt;) work(i); });
        for (auto &th : pool) th.join();
    }
    for (int i = 0; i < nt; i++) {
        const int off = (int)tree.nodes.size(), parent = tasks[i].parent;
        for (NDTree::Node &nd : sub[i].nodes) { for (int &c : nd.children) c += off; tree.nodes.push_back(std::move(nd)); }
        for (int r : sub[i].roots) { if (parent < 0) tree.roots.push_back(r + off); else tree.nodes[parent].children.push_back(r + off); }
    }
}

}  // namespace asam

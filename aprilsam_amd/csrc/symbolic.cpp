// symbolic.cpp — supernodal symbolic analysis on the pose-block graph (host, C++).
//
// Counterpart of the reference's symbolic stages: cs_schol (csparse.c:1693: etree, postorder, column
// counts), search_tree_create_from_smat (aprilsam.c:613-657: block elimination tree) and the idxs
// numbering (aprilsam.c:141-148) — restructured for a multifrontal GPU factorisation: every nested-
// dissection tree node is ONE dense front; we compute each front's update-row structure, the assembly
// tree, child->parent scatter maps, the factor->front assignment with deterministic gather lists, and a
// level schedule (fronts of one level are independent => one batched kernel launch).
#include "plan.h"
#include "solver.h"
#include "errors.h"

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <tuple>
#include <chrono>
#include <mutex>

namespace asam {

// Separator amalgamation (option amalg).  On a graph the size of M3500 nothing is bound by flops: the factorisation is a chain of
// dependent fronts, and a front on that chain costs about HAND = 12 us of hand-over (flag, extend-add of the children's blocks from
// memory, store + release of its update block) plus PERCOL = 0.19 us per own column (the in-register pivot chains;
// profiles/r05_front_times_m3500.txt), the back substitution another 2.7 us + 0.8 us per 64 columns.  A separator front that takes in a
// child SEPARATOR -- one dense front whose own part is the child separators first, then its own: a valid supernode, the blocks between two
// child separators are structural zeros treated as dense -- removes one hand-over from the path for the price of the child's columns on the
// parent's chain.  Bottom-up over the dissection tree with that model: cost(n) = hand + percol * own(n) + max over children cost(c); a node
// tries to take in its 1, 2, ... most expensive separator children (they sit on its critical path; a taken child brings the separators it
// took in itself, and its remaining children become n's) and keeps the prefix with the lowest cost(n), as long as the merged front still
// fits the single-workgroup kernel in panel mode (own columns x all rows in LDS; rows = own + the region's boundary, counted here) and owns
// at most amalg_max poses.  Leaves are never merged: they hold most of the poses and run side by side.  M3500: 9 levels -> 5-6.
static void amalgamate(NDTree &tree, int N, const std::vector<int> &ap, const std::vector<int> &ai, int max_own, size_t lds_budget) {
    const double HAND = 12.0 + 2.7, PERCOL = 0.19 + 0.8 / 64.0;         // us, up-sweep + down-sweep
    constexpr int REGION_MAX = 4096;                                      // subtrees with more poses are not looked at (their separators are far too wide to merge)
    const int nT = (int)tree.nodes.size();
    // post-order
    std::vector<int> order; order.reserve(nT);
    {
        std::vector<std::pair<int, size_t>> st;
        for (int r : tree.roots) {
            st.push_back({ r, 0 });
            while (!st.empty()) {
                auto &top = st.back();
                const NDTree::Node &nd = tree.nodes[top.first];
                if (top.second < nd.children.size()) { const int c = nd.children[top.second++]; st.push_back({ c, 0 }); continue; }
                order.push_back(top.first); st.pop_back();
            }
        }
    }
    std::vector<int> region(nT, 0);                                       // poses in the subtree
    for (int n : order) { region[n] = (int)tree.nodes[n].verts.size(); for (int c : tree.nodes[n].children) region[n] += region[c]; }
    // boundary of a subtree's region = update rows of its top front (the region is eliminated completely below it)
    std::vector<int> stamp_in(N, -1), stamp_b(N, -1), bnd(nT, -1);
    std::vector<int> stk, verts;
    auto boundary = [&](int n) {
        verts.clear(); stk.assign(1, n);
        while (!stk.empty()) { const int k = stk.back(); stk.pop_back(); for (int v : tree.nodes[k].verts) { stamp_in[v] = n; verts.push_back(v); } for (int c : tree.nodes[k].children) stk.push_back(c); }
        int cnt = 0;
        for (int v : verts) for (int e = ap[v]; e < ap[v + 1]; e++) { const int u = ai[e]; if (stamp_in[u] != n && stamp_b[u] != n) { stamp_b[u] = n; cnt++; } }
        return cnt;
    };
    std::vector<double> cost(nT, 0.0);
    std::vector<char> dead(nT, 0);
    for (int n : order) {
        NDTree::Node &nd = tree.nodes[n];
        auto eval = [&](size_t own, const std::vector<int> &kids) { double m = 0; for (int c : kids) m = std::max(m, cost[c]); return HAND + PERCOL * 3.0 * (double)own + m; };
        cost[n] = eval(nd.verts.size(), nd.children);
        if (nd.children.empty() || region[n] > REGION_MAX) continue;
        std::vector<int> seps;
        for (int c : nd.children) if (!tree.nodes[c].children.empty()) seps.push_back(c);
        if (seps.empty()) continue;
        std::sort(seps.begin(), seps.end(), [&](int a, int b) { return cost[a] != cost[b] ? cost[a] > cost[b] : a < b; });
        if (bnd[n] < 0) bnd[n] = boundary(n);
        size_t own = nd.verts.size();
        std::vector<int> kids = nd.children;
        double best = cost[n]; size_t best_k = 0;
        for (size_t k = 0; k < seps.size(); k++) {
            const NDTree::Node &ch = tree.nodes[seps[k]];
            own += ch.verts.size();
            const size_t R = 3 * (own + (size_t)bnd[n] + 1);
            if (own > (size_t)max_own || (R | 1) * 3 * own * 8 > lds_budget) break;
            kids.erase(std::find(kids.begin(), kids.end(), seps[k]));
            kids.insert(kids.end(), ch.children.begin(), ch.children.end());
            const double cst = eval(own, kids);
            if (cst < best - 1e-9) { best = cst; best_k = k + 1; }
        }
        if (!best_k) continue;
        std::vector<int> vs;
        for (size_t k = 0; k < best_k; k++) {
            NDTree::Node &ch = tree.nodes[seps[k]];
            vs.insert(vs.end(), ch.verts.begin(), ch.verts.end());
            nd.children.erase(std::find(nd.children.begin(), nd.children.end(), seps[k]));
            nd.children.insert(nd.children.end(), ch.children.begin(), ch.children.end());
            dead[seps[k]] = 1; ch.verts.clear(); ch.children.clear();
        }
        vs.insert(vs.end(), nd.verts.begin(), nd.verts.end());
        nd.verts.swap(vs);
        cost[n] = best;
    }
    std::vector<int> remap(nT, -1);
    int w = 0;
    for (int i = 0; i < nT; i++) if (!dead[i]) { remap[i] = w; if (w != i) tree.nodes[w] = std::move(tree.nodes[i]); w++; }
    tree.nodes.resize(w);
    for (auto &nd : tree.nodes) for (int &c : nd.children) c = remap[c];
    for (int &r : tree.roots) r = remap[r];
}

void build_plan(Plan &P, int N, int F, const int *fn, const double *xy, int leaf_nodes) {
    // one plan at a time: the dissection and the symbolic phases share ONE pool of planner threads (ordering.cpp: PlanPool) -- calls
    // from threads that drive different devices queue up here, everything else they do runs side by side
    static std::mutex build_mu;
    std::lock_guard<std::mutex> build_lock(build_mu);
    P = Plan();
    P.N = N; P.F = F; P.leaf_nodes = leaf_nodes;
    if (N <= 0) return;
    // APRILSAM_AMD_PLAN_PROFILE: wall-clock split of the planner phases on stderr
    const bool prof = getenv("APRILSAM_AMD_PLAN_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tph[8]; int nph = 0;
    auto stamp = [&] { if (prof && nph < 8) tph[nph++] = now(); };
    stamp();

    PlanSession session(N);              // the planner's worker threads stay awake for the length of this plan (ordering.cpp)
    // ---- pose adjacency (deduplicated CSR) ------------------------------------------------------------
    std::vector<int> ap(N + 1, 0), ai;
    {
        std::vector<int> deg(N, 0);
        for (int f = 0; f < F; f++) { int a = fn[2 * f], b = fn[2 * f + 1]; if (b >= 0 && a != b) { deg[a]++; deg[b]++; } }
        for (int i = 0; i < N; i++) ap[i + 1] = ap[i] + deg[i];
        ai.resize(ap[N]);
        std::vector<int> fill(ap.begin(), ap.end() - 1);
        for (int f = 0; f < F; f++) { int a = fn[2 * f], b = fn[2 * f + 1]; if (b >= 0 && a != b) { ai[fill[a]++] = b; ai[fill[b]++] = a; } }
        // sort + unique per row, then compact
        std::vector<int> np(N + 1, 0); int w = 0;
        for (int i = 0; i < N; i++) {
            std::sort(ai.begin() + ap[i], ai.begin() + ap[i + 1]);
            int last = -1;
            for (int e = ap[i]; e < ap[i + 1]; e++) if (ai[e] != last) { last = ai[e]; ai[w++] = last; }
            np[i + 1] = w;
        }
        ai.resize(w); ap.swap(np);
    }

    stamp();
    // ---- nested dissection tree, post-order numbering ----------------------------------------------
    NDTree tree;
    // Option pin_last = k ("recent poses last", the counterpart of the reference's constrained min-degree order,
    // aprilsam.c:1021-1098, which keeps the newest pose and the poses around its neighbours at the end of the elimination
    // order): the k highest pose ids are taken out of the dissection and form the ROOT front, so that a factor arriving at
    // the newest poses only touches the root.  Measured on the incremental demo: tools/inc_hist.py -> profiles/.
    const int pin = std::max(0, std::min(g_opt.pin_last, N - 1));
    if (pin > 0) {
        const int M = N - pin;
        std::vector<int> ap2(M + 1, 0), ai2;
        for (int i = 0; i < M; i++) { for (int e = ap[i]; e < ap[i + 1]; e++) if (ai[e] < M) ai2.push_back(ai[e]); ap2[i + 1] = (int)ai2.size(); }
        nested_dissection(M, ap2, ai2, xy, leaf_nodes, tree);
        NDTree::Node root;
        for (int v = M; v < N; v++) root.verts.push_back(v);
        root.children = tree.roots;
        tree.nodes.push_back(std::move(root));
        tree.roots.assign(1, (int)tree.nodes.size() - 1);
    } else
        nested_dissection(N, ap, ai, xy, leaf_nodes, tree);
    if (g_opt.amalg > 0) {          // (the LDS of a 16-wave workgroup's work lists comes off the budget: kernels.hip.h wl_bytes)
        const size_t wl = 64 * 24 + 16 * (128 * 8 + 64 * 8);
        amalgamate(tree, N, ap, ai, std::max(g_opt.amalg_max, 1), (size_t)std::max(g_opt.small_lds_kb, 32) * 1024 - wl);
    }
    const int nT = (int)tree.nodes.size();
    P.nF = nT;
    P.perm.assign(N, -1); P.pos.assign(N, -1);
    P.f_first.assign(nT, 0); P.f_nsb.assign(nT, 0);
    std::vector<int> tfront(nT, -1);        // ND tree node -> front index (post-order)
    {
        int nextpos = 0, nextfront = 0;
        std::vector<std::pair<int, size_t>> st;     // (tree node, next child)
        for (int r : tree.roots) {
            st.push_back({ r, 0 });
            while (!st.empty()) {
                auto &top = st.back();
                NDTree::Node &nd = tree.nodes[top.first];
                if (top.second < nd.children.size()) { int c = nd.children[top.second++]; st.push_back({ c, 0 }); continue; }
                int t = nextfront++;
                tfront[top.first] = t;
                P.f_first[t] = nextpos; P.f_nsb[t] = (int)nd.verts.size();
                for (int v : nd.verts) { P.perm[nextpos] = v; P.pos[v] = nextpos; nextpos++; }
                st.pop_back();
            }
        }
        assert(nextpos == N && nextfront == nT);
    }
    std::vector<int> pos_front(N);
    for (int t = 0; t < nT; t++) for (int k = 0; k < P.f_nsb[t]; k++) pos_front[P.f_first[t] + k] = t;

    stamp();
    // ---- structure of every front (block positions > own range), assembly tree -----------------------
    P.f_nub.assign(nT, 0); P.f_parent.assign(nT, -1); P.f_rows_ptr.assign(nT + 1, 0);
    std::vector<std::vector<int>> kids(nT);
    std::vector<int> mark(N, -1), tmp;
    std::vector<std::vector<int>> strct(nT);
    for (int t = 0; t < nT; t++) {
        const int last = P.f_first[t] + P.f_nsb[t] - 1;
        tmp.clear();
        for (int p = P.f_first[t]; p <= last; p++) {
            int v = P.perm[p];
            for (int e = ap[v]; e < ap[v + 1]; e++) { int q = P.pos[ai[e]]; if (q > last && mark[q] != t) { mark[q] = t; tmp.push_back(q); } }
        }
        for (int c : kids[t]) for (int q : strct[c]) if (q > last && mark[q] != t) { mark[q] = t; tmp.push_back(q); }
        std::sort(tmp.begin(), tmp.end());
        strct[t] = tmp;
        P.f_nub[t] = (int)tmp.size();
        if (!tmp.empty()) { int par = pos_front[tmp[0]]; assert(par > t); P.f_parent[t] = par; kids[par].push_back(t); }
    }
    for (int t = 0; t < nT; t++) P.f_rows_ptr[t + 1] = P.f_rows_ptr[t] + P.f_nub[t];
    P.f_rows.resize(P.f_rows_ptr[nT]); P.f_rel.assign(P.f_rows_ptr[nT], -1);
    for (int t = 0; t < nT; t++) std::copy(strct[t].begin(), strct[t].end(), P.f_rows.begin() + P.f_rows_ptr[t]);

    auto local_index = [&](int t, int p) -> int {      // block index of position p inside front t (or -1)
        if (p >= P.f_first[t] && p < P.f_first[t] + P.f_nsb[t]) return p - P.f_first[t];
        const int *b = P.f_rows.data() + P.f_rows_ptr[t], *e = b + P.f_nub[t];
        const int *it = std::lower_bound(b, e, p);
        if (it == e || *it != p) return -1;
        return P.f_nsb[t] + (int)(it - b);
    };
    for (int t = 0; t < nT; t++) {
        int par = P.f_parent[t];
        for (int64_t k = P.f_rows_ptr[t]; k < P.f_rows_ptr[t + 1]; k++) {
            int li = local_index(par, P.f_rows[k]);
            if (li < 0) fail(ERR_INTERNAL, "symbolic inconsistency (front %d row %d)", t, P.f_rows[k]);
            P.f_rel[k] = li;
        }
    }
    P.ch_ptr.assign(nT + 1, 0);
    for (int t = 0; t < nT; t++) P.ch_ptr[t + 1] = P.ch_ptr[t] + (int)kids[t].size();
    P.ch_idx.resize(P.ch_ptr[nT]);
    for (int t = 0; t < nT; t++) std::copy(kids[t].begin(), kids[t].end(), P.ch_idx.begin() + P.ch_ptr[t]);

    stamp();
    // ---- levels ------------------------------------------------------------------------------------------
    P.f_level.assign(nT, 0);
    for (int t = 0; t < nT; t++) { int par = P.f_parent[t]; if (par >= 0) P.f_level[par] = std::max(P.f_level[par], P.f_level[t] + 1); }
    P.nLevels = 0;
    for (int t = 0; t < nT; t++) P.nLevels = std::max(P.nLevels, P.f_level[t] + 1);
    P.lev_ptr.assign(P.nLevels + 1, 0);
    for (int t = 0; t < nT; t++) P.lev_ptr[P.f_level[t] + 1]++;
    for (int l = 0; l < P.nLevels; l++) P.lev_ptr[l + 1] += P.lev_ptr[l];
    P.lev_fronts.resize(nT);
    { std::vector<int> fill(P.lev_ptr.begin(), P.lev_ptr.end() - 1); for (int t = 0; t < nT; t++) P.lev_fronts[fill[P.f_level[t]]++] = t; }

    stamp();
    // ---- factor -> front assignment + deterministic gather lists -------------------------------------
    P.fac_front.assign(F, -1); P.fac_la.assign(F, -1); P.fac_lb.assign(F, -1); P.fac_swap.assign(F, 0);
    std::atomic<int> lost{ -1 };
    plan_parallel_for(F, 4096, [&](int f0, int f1) {               // the searches, side by side ...
        for (int f = f0; f < f1; f++) {
            int a = fn[2 * f], b = fn[2 * f + 1];
            if (a < 0 || a >= N || b >= N || a == b) continue;          // malformed: ignored
            int pa = P.pos[a], pb = b >= 0 ? P.pos[b] : -1;
            int t = pos_front[(pb >= 0 && pb < pa) ? pb : pa];
            int la = local_index(t, pa), lb = pb >= 0 ? local_index(t, pb) : -1;
            if (la < 0 || (pb >= 0 && lb < 0)) { lost.store(f); continue; }
            P.fac_front[f] = t; P.fac_la[f] = la; P.fac_lb[f] = lb;
        }
    });
    if (lost.load() >= 0) fail(ERR_INTERNAL, "factor %d not inside its front", lost.load());
    for (int f = 0; f < F; f++) if (P.fac_front[f] >= 0 && P.fac_lb[f] >= 0) P.fac_swap[f] = P.fac_la[f] < P.fac_lb[f];
    // packed sort keys below: 16 bits of local block column, 17 of local block row + 1, 30 of contribution id, 1 of kind.  (Round 6: the id had
    // 26 bits -- 22 million factors -- which a 2 500 x 2 500 lattice exceeds while its 190 GB of fronts still fit one MI355X.)
    constexpr int KEY_SRC_BITS = 30, KEY_ROW_BITS = 17, KEY_COL_BITS = 16;
    static_assert(1 + KEY_SRC_BITS + KEY_ROW_BITS + KEY_COL_BITS == 64 && sizeof(unsigned long long) == 8, "packed sort keys");
    if ((size_t)3 * F >= (1u << KEY_SRC_BITS) || (size_t)5 * F >= 0x7fffffffu)
        fail(ERR_UNSUPPORTED, "%d factors: this build handles up to %u (packed sort keys of the symbolic analysis)", F, (1u << KEY_SRC_BITS) / 3 - 1);
    for (int t = 0; t < nT; t++)
        if (P.f_nsb[t] + P.f_nub[t] + 1 >= (1 << KEY_COL_BITS))
            fail(ERR_UNSUPPORTED, "front %d has %d block rows: this build handles up to %d (packed sort keys of the symbolic analysis)", t, P.f_nsb[t] + P.f_nub[t] + 1, (1 << KEY_COL_BITS) - 1);
    P.bd_front_ptr.clear(); P.rd_front_ptr.clear();       // (the per-kind gather lists are derived on demand: build_gather_lists)

    stamp();
    // ---- unified destination records + contribution slots (device layout) --------------------------------
    {
        // One record per destination block (brow -1 = the rhs row) with the consecutive slots of its contributions.  Order inside a
        // front: (col, row with the rhs row first, contribution id) -- a counting sort by front, then a sort of 64-bit packed keys
        // inside every (small) bucket, the buckets side by side: several times faster than one comparison sort over all 5 F
        // entries, and planning sits on the cold-call path.
        std::vector<int> ptr(nT + 1, 0);
        for (int f = 0; f < F; f++) if (P.fac_front[f] >= 0) ptr[P.fac_front[f] + 1] += P.fac_lb[f] >= 0 ? 5 : 2;
        for (int t = 0; t < nT; t++) ptr[t + 1] += ptr[t];
        const size_t total = (size_t)ptr[nT];
        std::vector<unsigned long long> key(total);
        auto pack = [](int col, int row, int src, int rhs) {
            return ((unsigned long long)(unsigned)col << (1 + KEY_SRC_BITS + KEY_ROW_BITS)) | ((unsigned long long)(unsigned)(row + 1) << (1 + KEY_SRC_BITS)) | ((unsigned long long)(unsigned)src << 1) | (unsigned)rhs;
        };
        {
            std::vector<int> fill(ptr.begin(), ptr.end() - 1);
            for (int f = 0; f < F; f++) {
                const int t = P.fac_front[f], la = P.fac_la[f], lb = P.fac_lb[f];
                if (t < 0) continue;
                int &w = fill[t];
                key[w++] = pack(la, la, 3 * f + 0, 0); key[w++] = pack(la, -1, 2 * f + 0, 1);
                if (lb >= 0) {
                    key[w++] = pack(std::min(la, lb), std::max(la, lb), 3 * f + 1, 0);
                    key[w++] = pack(lb, lb, 3 * f + 2, 0); key[w++] = pack(lb, -1, 2 * f + 1, 1);
                }
            }
        }
        auto same_dest = [](unsigned long long a, unsigned long long b) { return (a >> (1 + KEY_SRC_BITS)) == (b >> (1 + KEY_SRC_BITS)); };     // same (col, row)
        P.dest_front_ptr.assign(nT + 1, 0);
        plan_parallel_for(nT, 64, [&](int t0, int t1) {
            for (int t = t0; t < t1; t++) {
                std::sort(key.begin() + ptr[t], key.begin() + ptr[t + 1]);
                int fresh = 0;
                for (int i = ptr[t]; i < ptr[t + 1]; i++) fresh += i == ptr[t] || !same_dest(key[i], key[i - 1]);
                P.dest_front_ptr[t + 1] = fresh;
            }
        });
        for (int t = 0; t < nT; t++) P.dest_front_ptr[t + 1] += P.dest_front_ptr[t];
        P.dest.resize((size_t)P.dest_front_ptr[nT]);
        P.slot_blk.assign((size_t)3 * F, -1); P.slot_rhs.assign((size_t)2 * F, -1);
        plan_parallel_for(nT, 64, [&](int t0, int t1) {
            for (int t = t0; t < t1; t++) {
                int d = P.dest_front_ptr[t] - 1;
                for (int i = ptr[t]; i < ptr[t + 1]; i++) {
                    const unsigned long long k = key[i];
                    const int col = (int)(k >> (1 + KEY_SRC_BITS + KEY_ROW_BITS)), row = (int)((k >> (1 + KEY_SRC_BITS)) & ((1u << KEY_ROW_BITS) - 1)) - 1,
                              src = (int)((k >> 1) & ((1u << KEY_SRC_BITS) - 1)), rhs = (int)(k & 1);
                    if (i == ptr[t] || !same_dest(k, key[i - 1])) P.dest[++d] = { row, col, i, i };
                    P.dest[d].src_end = i + 1;
                    (rhs ? P.slot_rhs : P.slot_blk)[src] = i;
                }
            }
        });
        P.n_slots = (int)total;
    }

    stamp();
    // ---- HBM pool offsets + statistics -----------------------------------------------------------------
    P.f_off.assign(nT, 0);
    int64_t off = 0;
    P.max_rows = 0; P.nnzL = 0; P.flops = 0;
    for (int t = 0; t < nT; t++) {
        P.f_off[t] = off;
        int64_t sz = (int64_t)P.rows(t) * P.cols(t);
        off += (sz + 31) & ~int64_t(31);                          // 256-byte aligned fronts
        int ns = 3 * P.f_nsb[t], nu = 3 * P.f_nub[t];
        P.max_rows = std::max(P.max_rows, ns + nu);
        P.nnzL += (int64_t)ns * (ns + 1) / 2 + (int64_t)nu * ns;
        for (int q = 0; q < ns; q++) { double c = (double)(ns - q) + nu; P.flops += c * c; }
    }
    P.pool_doubles = off;
    stamp();
    if (prof && nph == 8)
        fprintf(stderr, "aprilsam_amd planner N=%d F=%d: adjacency %.3f dissection %.3f front structure %.3f levels %.3f factor lists %.3f records+slots %.3f offsets %.3f | total %.3f ms\n",
                N, F, tph[1] - tph[0], tph[2] - tph[1], tph[3] - tph[2], tph[4] - tph[3], tph[5] - tph[4], tph[6] - tph[5], tph[7] - tph[6], tph[7] - tph[0]);
}

// The per-kind gather lists (block destinations / rhs destinations with their contributing factors in a fixed order): a
// restatement of the destination records for the host-side emulator of the tests (aprilsam_amd_plan_query "bd_*" / "rd_*");
// the device never reads them, so they are derived on demand.
void build_gather_lists(Plan &P) {
    if (!P.bd_front_ptr.empty()) return;
    const int F = P.F, nT = P.nF;
    struct Dest { int front, col, row, src; };
    std::vector<Dest> bd, rd; bd.reserve((size_t)3 * F); rd.reserve((size_t)2 * F);
    for (int f = 0; f < F; f++) {
        const int t = P.fac_front[f], la = P.fac_la[f], lb = P.fac_lb[f];
        if (t < 0) continue;
        bd.push_back({ t, la, la, 3 * f + 0 });
        rd.push_back({ t, la, 0, 2 * f + 0 });
        if (lb >= 0) {
            bd.push_back({ t, std::min(la, lb), std::max(la, lb), 3 * f + 1 });
            bd.push_back({ t, lb, lb, 3 * f + 2 });
            rd.push_back({ t, lb, 0, 2 * f + 1 });
        }
    }
    auto sort_dests = [&](std::vector<Dest> &v) {
        std::vector<int> ptr(nT + 1, 0);
        for (const Dest &d : v) ptr[d.front + 1]++;
        for (int t = 0; t < nT; t++) ptr[t + 1] += ptr[t];
        std::vector<unsigned long long> key(v.size());
        { std::vector<int> fill(ptr.begin(), ptr.end() - 1);
          for (const Dest &d : v) key[fill[d.front]++] = ((unsigned long long)(unsigned)d.col << 44) | ((unsigned long long)(unsigned)(d.row + 1) << 26) | (unsigned)d.src; }
        for (int t = 0; t < nT; t++) {
            std::sort(key.begin() + ptr[t], key.begin() + ptr[t + 1]);
            for (int i = ptr[t]; i < ptr[t + 1]; i++) v[i] = { t, (int)(key[i] >> 44), (int)((key[i] >> 26) & 0x3ffff) - 1, (int)(key[i] & 0x3ffffff) };
        }
    };
    sort_dests(bd);
    sort_dests(rd);
    auto compress = [&](const std::vector<Dest> &v, std::vector<int> &front_ptr, std::vector<int> &row, std::vector<int> &col,
                        std::vector<int> &src_ptr, std::vector<int> &src) {
        front_ptr.assign(nT + 1, 0); row.clear(); col.clear(); src_ptr.clear(); src.clear();
        src_ptr.push_back(0);
        for (size_t i = 0; i < v.size(); i++) {
            bool fresh = i == 0 || v[i].front != v[i - 1].front || v[i].col != v[i - 1].col || v[i].row != v[i - 1].row;
            if (fresh) { if (i) src_ptr.push_back((int)src.size()); row.push_back(v[i].row); col.push_back(v[i].col); front_ptr[v[i].front + 1]++; }
            src.push_back(v[i].src);
        }
        if (!v.empty()) src_ptr.push_back((int)src.size());
        for (int t = 0; t < nT; t++) front_ptr[t + 1] += front_ptr[t];
    };
    std::vector<int> dummy_row;
    compress(bd, P.bd_front_ptr, P.bd_row, P.bd_col, P.bd_src_ptr, P.bd_src);
    compress(rd, P.rd_front_ptr, dummy_row, P.rd_col, P.rd_src_ptr, P.rd_src);
}

}  // namespace asam

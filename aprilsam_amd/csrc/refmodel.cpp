// refmodel.cpp — host-side model of the reference's BOOKKEEPING for the incremental path.
//
// Measured on this build's oracle (tools/inc_experiment in DESIGN.md §7): on the poses it touches, the
// reference's incremental result equals the exact solution of the incremental linear system to 1e-12.
// What makes april_graph_cholesky_inc differ from "solve and update everything" is bookkeeping only:
//   * which poses are visited / updated by solve_node (aprilsam.c:721-779): all of them when more than 5
//     tree nodes are affected, otherwise the marked root paths plus their direct children (x computed,
//     delta_X overwritten, not updated);
//   * the relinearisation counter start_over (aprilsam.c:741-751) that triggers the batch fall-back (:566).
// Both are functions of the reference's block elimination tree (aprilsam.c:613-657, 908-987), i.e. of ITS
// elimination order (aprilsam.c:999-1249) extended by identity for new poses (:393-396).  This file restates
// that order (including the quirks that decide ties) and the tree as pure integer logic; the numbers come
// from the GPU.  Nothing here touches floating-point data except the threshold tests on x.
#include "refmodel.h"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <unordered_map>

namespace asam {

namespace {

// binary max-heap with the sift rules of common/zmaxheap.c:134-159 (add: climb while parent < v) and
// :181-241 (remove root: last element to the top, descend towards the larger child, left on ties)
struct RefHeap {
    std::vector<float> val; std::vector<int> item;
    void add(int it, float v) {
        int idx = (int)val.size();
        val.push_back(v); item.push_back(it);
        while (idx > 0) {
            int parent = (idx - 1) / 2;
            if (val[parent] >= v) break;
            std::swap(val[idx], val[parent]); std::swap(item[idx], item[parent]);
            idx = parent;
        }
    }
    bool pop(int *it, float *v) {
        if (val.empty()) return false;
        *it = item[0]; *v = val[0];
        int n = (int)val.size() - 1;
        if (n == 0) { val.clear(); item.clear(); return true; }
        val[0] = val[n]; item[0] = item[n];
        val.pop_back(); item.pop_back();
        int parent = 0;
        const float ps = val[0];
        while (parent < n) {
            int left = 2 * parent + 1, right = left + 1;
            float ls = left < n ? val[left] : -INFINITY, rs = right < n ? val[right] : -INFINITY;
            if (ps >= ls && ps >= rs) break;
            int ch = (ls >= rs) ? left : right;
            std::swap(val[parent], val[ch]); std::swap(item[parent], item[ch]);
            parent = ch;
        }
        return true;
    }
};

}  // namespace

// aprilsam.c:999-1249 restated.  adj: sorted neighbour lists without self loops (the rows of the symbolic
// matrix built at aprilsam.c:104-114).  Returns position -> node.
std::vector<int> ref_min_degree_order(int N, const std::vector<std::vector<int>> &adj) {
    // The reference keeps the elimination graph explicitly (every eliminated pose turns its neighbours into a clique, :1150-1222) but
    // only ever looks at the SIZE of a pose's neighbour list, when the pose comes off a list.  Here the graph is a quotient graph: an
    // eliminated pose becomes an element holding its neighbours at that moment and absorbs the elements it touched; a pose's degree
    // -- the same number -- is counted when it is asked for, from its original neighbours and the elements it belongs to.  O(d) per
    // elimination instead of O(d^2), and no vector per pose that grows with the fill: half the cost of this model after a batch step
    // (M3500: 3.4 -> 1.75 ms in the build container), which every fall-back step of an incremental run pays under the GPU's work --
    // the demo's 49 fall-backs 94 -> 75 ms.  tests/test_refmodel.py holds the orders against the reference's.
    const std::vector<std::vector<int>> &nb = adj;      // (phase 1 below: original neighbour lists)
    // (both in flat arenas: ten thousand small vectors cost more than everything they hold)
    std::vector<int> el_buf; std::vector<int> el_off, el_len; std::vector<char> absorbed;      // element e: poses el_buf[el_off[e] .. + el_len[e])
    std::vector<int> eb; std::vector<int> e_off(N, 0), e_len(N, 0), e_cap(N, 0);                  // pose i: elements eb[e_off[i] .. + e_len[i])
    el_buf.reserve((size_t)8 * N); eb.reserve((size_t)8 * N); el_off.reserve(N); el_len.reserve(N); absorbed.reserve(N);
    auto e_push = [&](int i, int e) {
        if (e_len[i] == e_cap[i]) {                         // move the list to the end of the arena with twice the room
            const int ncap = e_cap[i] ? 2 * e_cap[i] : 4, noff = (int)eb.size();
            eb.resize(eb.size() + (size_t)ncap);
            for (int k = 0; k < e_len[i]; k++) eb[noff + k] = eb[e_off[i] + k];
            e_off[i] = noff; e_cap[i] = ncap;
        }
        eb[e_off[i] + e_len[i]++] = e;
    };
    std::vector<int> mark(N, 0); int tok = 0;
    // the reference's lists are FIFOs (zarray remove at 0, add at the end) and a pose may sit in several of them at once (phase 1
    // below adds the unmarked ones again and again): entries {pose, next} in one arena, a head and a tail per list -- no allocation
    // per list (there are thousands: a list that is re-created for a key without a hash put is a new one)
    std::vector<int> ent_node, ent_next, l_head, l_tail;
    ent_node.reserve((size_t)4 * N); ent_next.reserve((size_t)4 * N); l_head.reserve(N); l_tail.reserve(N);
    auto new_list = [&]() { l_head.push_back(-1); l_tail.push_back(-1); return (int)l_head.size() - 1; };
    auto list_push = [&](int l, int node) {
        const int e = (int)ent_node.size();
        ent_node.push_back(node); ent_next.push_back(-1);
        if (l_tail[l] >= 0) ent_next[l_tail[l]] = e; else l_head[l] = e;
        l_tail[l] = e;
    };
    auto list_pop = [&](int l) {                         // (the list is not empty)
        const int e = l_head[l], node = ent_node[e];
        l_head[l] = ent_next[e];
        if (l_head[l] < 0) l_tail[l] = -1;
        return node;
    };
    // key -> list (only lists created with a hash put).  Keys are degrees plus at most 2 (N - 1) (phase 1): a table, not a hash map
    size_t maxdeg = 0; for (int i = 0; i < N; i++) maxdeg = std::max(maxdeg, adj[i].size());
    std::vector<int> registry((size_t)3 * N + maxdeg + 8, -1);
    RefHeap heap;
    auto add_registered = [&](uint32_t key, int node) {
        if (registry[key] >= 0) { list_push(registry[key], node); return; }
        const int l = new_list(); list_push(l, node);
        registry[key] = l;
        heap.add(l, (float)(-1.0 * key));
    };
    std::vector<char> set_marker(N, 0);
    if (N > 0) {
        // "most recent node set the lowest score" (:1021-1098)
        const int rowi = N - 1;
        add_registered((uint32_t)(nb[rowi].size() + 2 * rowi), rowi);
        set_marker[rowi] = 1;
        for (size_t i = 0; i < nb[rowi].size(); i++) {
            const int choose = nb[rowi][i];
            for (int idx = choose - 5; idx < choose + 5; idx++) {
                if (idx < 0 || idx > N - 1) continue;
                if (set_marker[idx]) continue;
                add_registered((uint32_t)(nb[idx].size() + rowi), idx);
                set_marker[idx] = 1;
                // :1080-1095 — the loop variable itself is used as a node id and is never marked
                for (int j = 0; j < (int)nb[idx].size(); j++) {
                    if (set_marker[j]) continue;
                    add_registered((uint32_t)(nb[j].size() + rowi), j);
                }
            }
        }
    }
    for (int rowi = 0; rowi < N - 1; rowi++) {           // :1100-1116
        if (set_marker[rowi]) continue;
        add_registered((uint32_t)nb[rowi].size(), rowi);
    }
    std::vector<int> ordering; ordering.reserve(N);
    std::vector<char> gone(N, 0);
    // (the membership tests below are data dependent coin flips: written without branches -- a pose that is gone may be stamped as
    // well, it is never counted -- the model after a batch step of M3500 took 1.75 ms with them, two thirds of it mispredictions)
    auto degree = [&](int i) {
        tok++; mark[i] = tok;
        int cnt = 0;
        for (int v : adj[i]) { cnt += ((int)!gone[v] & (int)(mark[v] != tok)); mark[v] = tok; }
        int w = 0;
        for (int k = 0; k < e_len[i]; k++) {
            const int e = eb[e_off[i] + k];
            if (absorbed[e]) continue;
            eb[e_off[i] + w++] = e;
            const int *L = el_buf.data() + el_off[e];
            const int len = el_len[e];
            for (int q = 0; q < len; q++) { const int v = L[q]; cnt += ((int)!gone[v] & (int)(mark[v] != tok)); mark[v] = tok; }
        }
        e_len[i] = w;
        return cnt;
    };
    auto eliminate = [&](int b) {
        gone[b] = 1;
        tok++; mark[b] = tok;
        const int off = (int)el_buf.size();
        // room for every candidate first, then unconditional stores with a conditional advance (same members, same order)
        size_t room = adj[b].size();
        for (int k = 0; k < e_len[b]; k++) { const int e = eb[e_off[b] + k]; room += absorbed[e] ? 0 : (size_t)el_len[e]; }
        el_buf.resize((size_t)off + room);
        int *out = el_buf.data() + off; int n = 0;
        for (int v : adj[b]) { out[n] = v; n += ((int)!gone[v] & (int)(mark[v] != tok)); mark[v] = tok; }
        for (int k = 0; k < e_len[b]; k++) {
            const int e = eb[e_off[b] + k];
            if (absorbed[e]) continue;
            const int *L = el_buf.data() + el_off[e];       // (el_buf does not move any more: resized above)
            const int len = el_len[e];
            for (int q = 0; q < len; q++) { const int v = L[q]; out[n] = v; n += ((int)!gone[v] & (int)(mark[v] != tok)); mark[v] = tok; }
            absorbed[e] = 1;
        }
        el_buf.resize((size_t)off + (size_t)n);
        const int ne = (int)el_off.size(), len = n;
        el_off.push_back(off); el_len.push_back(len); absorbed.push_back(0);
        for (int q = 0; q < len; q++) e_push(el_buf[off + q], ne);
    };
    int li; float v;
    while (heap.pop(&li, &v)) {                           // :1128-1238
        while (l_head[li] >= 0) {
            const int b = list_pop(li);
            if (gone[b]) continue;
            const int deg = degree(b);
            if ((float)deg <= -v) { ordering.push_back(b); eliminate(b); }      // its neighbours become a clique (:1150-1222)
            else {                                                      // :1224-1235 (no hash put for a new list)
                const uint32_t key = (uint32_t)deg;
                if (registry[key] >= 0) list_push(registry[key], b);
                else { const int l = new_list(); list_push(l, b); heap.add(l, (float)(-1.0 * key)); }
            }
        }
    }
    return ordering;
}

// block elimination tree under `ord` (position -> node): parent[c] = node owning the first off-diagonal block of
// c's row in U (aprilsam.c:635-651) = Liu's elimination tree of the pose graph, -1 when the row has none
static void block_etree(int N, const std::vector<std::vector<int>> &adj, const std::vector<int> &ord, const std::vector<int> &pos,
                        std::vector<int> &parent) {
    parent.assign(N, -1);
    std::vector<int> anc(N, -1);               // by position
    std::vector<int> ppos(N, -1);
    for (int p = 0; p < N; p++) {
        const int v = ord[p];
        for (int w : adj[v]) {
            int q = pos[w];
            if (q >= p) continue;
            while (q != -1 && q < p) {
                int next = anc[q];
                anc[q] = p;
                if (next == -1) ppos[q] = p;
                q = next;
            }
        }
    }
    for (int p = 0; p < N; p++) parent[ord[p]] = ppos[p] >= 0 ? ord[ppos[p]] : -1;
}

void RefModel::add_factor_edges(int a, int b) {
    if (b < 0 || a == b) return;
    auto ins = [&](int u, int w) { auto &L = adj[u]; auto it = std::lower_bound(L.begin(), L.end(), w); if (it == L.end() || *it != w) L.insert(it, w); };
    ins(a, b); ins(b, a);
}

void RefModel::set_parent(int v, int p) {
    const int old = parent[v];
    if (old == p) return;
    if (old >= 0) { auto &k = kids[old]; k.erase(std::find(k.begin(), k.end(), v)); }
    parent[v] = p;
    if (p >= 0) kids[p].push_back(v);
}

// The elimination tree after one more edge (u, v), without recomputing it: the root paths of the two poses merge into one
// chain, in elimination order, up to their first common ancestor (Liu's path merging).  a is the earlier pose, b must
// become its ancestor: climb from a while the ancestors still precede b, splice b in where they stop, then go on with b and
// the displaced ancestor.  Every step moves up a root path, so an insertion costs the length of the two paths -- the full
// recomputation it replaces (block_etree, all N poses) was a third of an incremental step's host time on M3500.
void RefModel::insert_edge(int u, int v) {
    if (v < 0 || u == v) return;
    int a = pos[u] < pos[v] ? u : v, b = pos[u] < pos[v] ? v : u;
    for (;;) {
        const int p = parent[a];
        if (p == b) break;
        if (p == -1) { set_parent(a, b); break; }
        if (pos[p] < pos[b]) a = p;
        else { set_parent(a, b); a = b; b = p; }
    }
}

// after a batch step on the first n_nodes nodes / n_factors factors (aprilsam.c:121,269)
void RefModel::batch(int n_nodes, int n_factors, const int *fa, const int *fb) {
    N = n_nodes; F = n_factors;
    auto nowt = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double q0 = nowt();
    adj.assign(N, {});
    for (int f = 0; f < F; f++) add_factor_edges(fa[f], fb[f]);
    double q1 = nowt();
    ord = ref_min_degree_order(N, adj);
    double q2 = nowt();
    pos.assign(N, -1);
    for (int p = 0; p < N; p++) pos[ord[p]] = p;
    block_etree(N, adj, ord, pos, parent);
    double q3 = nowt();
    kids.assign(N, {});
    for (int i = 0; i < N; i++) if (parent[i] >= 0) kids[parent[i]].push_back(i);
    if (getenv("APRILSAM_AMD_PLAN_PROFILE")) fprintf(stderr, "aprilsam_amd reference model N=%d: adjacency %.3f min-degree order %.3f tree %.3f children %.3f ms\n", N, q1 - q0, q2 - q1, q3 - q2, nowt() - q3);
    changed.assign(N, 0); relin.assign(N, 0);
    start_over = 0; naffected = 0;
    root = N > 0 ? ord[N - 1] : -1;
    valid = true;
}

// aprilsam.c:393-498 + 550: extend the order by identity, mark the root paths of the new factors' poses in
// the OLD tree, then re-parent (the structure of U after the partial re-factorisation)
void RefModel::inc_begin(int n_nodes, int n_factors, const int *fa, const int *fb) {
    const int oldN = N;
    adj.resize(n_nodes); parent.resize(n_nodes, -1); changed.resize(n_nodes, 0); relin.resize(n_nodes, 0); kids.resize(n_nodes);
    for (int i = oldN; i < n_nodes; i++) { ord.push_back(i); pos.push_back(i); }
    naffected = 0; old_old_cross = 0;
    // A new factor between two poses that both predate this call, neither an ancestor of the other in the tree as it stands
    // BEFORE the call: the reference's partial re-factorisation walks that old tree children first (aprilsam.c:850-906) and
    // finalises one pose's rows before the other's have been added to them -- its result is not the solution of its own normal
    // equations there (tests/test_gpu_api_surface.py pins the distance).  Counted so that the caller can be told.
    for (int f = F; f < n_factors; f++) {
        const int u = fa[f], v = fb[f];
        if (v < 0 || u == v || u >= oldN || v >= oldN) continue;
        const int lo = pos[u] < pos[v] ? u : v, hi = pos[u] < pos[v] ? v : u;
        int a = lo;
        while (a != -1 && a != hi && pos[a] < pos[hi]) a = parent[a];
        if (a != hi) old_old_cross++;
    }
    for (int f = F; f < n_factors; f++) {
        const int nodes[2] = { fa[f], fb[f] };
        for (int z0 = 0; z0 < (fb[f] >= 0 ? 2 : 1); z0++) {
            int n = nodes[z0];
            while (!changed[n]) {
                changed[n] = 1; naffected++;
                if (parent[n] != -1) n = parent[n]; else break;
            }
        }
    }
    for (int f = F; f < n_factors; f++) { add_factor_edges(fa[f], fb[f]); insert_edge(fa[f], fb[f]); }
    N = n_nodes; F = n_factors;
    root = ord[N - 1];
}

// the incrementally maintained tree against a full recomputation (tests): 0 = identical
int RefModel::check_tree() const {
    std::vector<int> full;
    block_etree(N, adj, ord, pos, full);
    for (int i = 0; i < N; i++) if (full[i] != parent[i]) return i + 1;
    std::vector<int> cnt(N, 0);
    for (int i = 0; i < N; i++) { for (int k : kids[i]) { if (parent[k] != i) return -(i + 1); cnt[k]++; } }
    for (int i = 0; i < N; i++) if (cnt[i] != (parent[i] >= 0 ? 1 : 0)) return -(N + i + 1);
    return 0;
}

// aprilsam.c:721-779, split in two: the traversal is purely structural (labels + tree), so it can be planned
// BEFORE the numbers exist — the GPU then only back-substitutes the fronts that hold a visited pose.
// plan_visit: poses solve_node touches, in order; update == true -> node->update(x) (state, delta_X),
// false -> only delta_X = x (an unmarked child reached when naffected <= 5; the walk stops there).
void RefModel::plan_visit(std::vector<Visit> &out) {
    out.clear();
    std::vector<int> &stack = visit_stack; stack.clear(); stack.push_back(root);
    while (!stack.empty()) {
        const int n = stack.back(); stack.pop_back();
        bool update = true;
        if (naffected > 5) changed[n] = 0;
        else if (changed[n] == 1) changed[n] = 0;
        else update = false;
        out.push_back({ n, update });
        if (!update) continue;                       // :769 returns before the children
        for (int k : kids[n]) stack.push_back(k);
    }
}
// relinearisation counter over the visited poses (aprilsam.c:741-751)
void RefModel::count_relinearized(const double *x, double dxy, double dth, const std::vector<Visit> &visits) {
    for (const Visit &v : visits) {
        const double *xi = x + 3 * (size_t)v.node;
        if (std::fabs(xi[0]) > dxy || std::fabs(xi[1]) > dxy || std::fabs(xi[2]) > dth) {
            if (!relin[v.node]) { relin[v.node] = 1; start_over++; }
        }
    }
}
void RefModel::solve_visit(const double *x, double dxy, double dth, const std::function<void(int, bool)> &visit) {
    std::vector<Visit> v;
    plan_visit(v);
    count_relinearized(x, dxy, dth, v);
    for (const Visit &w : v) visit(w.node, w.update);
}

}  // namespace asam

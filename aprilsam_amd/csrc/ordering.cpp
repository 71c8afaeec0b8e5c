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
#include "errors.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <numeric>
#include <string>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>

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

// Worker threads of the planner, created once per process and parked between plans (creating up to 16 threads per plan
// cost about a tenth of a cold M3500 plan).  run(n, fn): fn(0..n-1) on the workers AND the caller, back when all are done;
// the caller alone finishes the job if no worker ever shows up (a forked child has none).  After a job a worker spins for
// a short while before it sleeps: inside one plan the jobs follow each other within tens of microseconds.
// (no PAUSE in the wait loops: under a hypervisor with pause-loop exiting a spinning virtual CPU is taken off its core for a
// whole time slice, which turns a wait of microseconds into one of hundreds; the waits here are short and bounded)
static inline void cpu_relax() { asm volatile("" ::: "memory"); }
class PlanPool {
    // Two job slots used alternately (job number g lives in slot g & 1).  Picking a job up takes no lock -- a mutex handed
    // from worker to worker costs a futex wake per hand-over, which on a virtual machine is longer than the jobs -- so a
    // slot is guarded Dekker-style: a worker announces "inside job g" (state) and then checks the slot's tag; the caller
    // closes the slot (tag = -1), then waits for every "inside" announcement of its previous use to be withdrawn, then refills it.
    struct alignas(128) Job { std::function<void(int, int)> fn; int n = 0; std::atomic<int> next{ 0 }, done{ 0 }, failed{ 0 }; std::atomic<long long> tag{ -1 }; };
    struct alignas(128) WState { std::atomic<long long> v{ 0 }; };      // 2 g + 1: inside job g; even: not inside any
    Job slot[2];
    std::unique_ptr<WState[]> state;
    std::atomic<long long> gen{ 0 };
    std::mutex mu; std::condition_variable cv; std::atomic<int> sleepers{ 0 };     // only for workers idle long enough to go to sleep
    std::atomic<int> sessions{ 0 };            // plans in progress: their jobs follow each other within microseconds, workers stay awake
    std::mutex one_caller;                     // (plans of different params may be built on different application threads)
    int nworkers = 0;
    static void drain(Job &j, int who) { for (int i; (i = j.next.fetch_add(1, std::memory_order_acq_rel)) < j.n;) {
        try { j.fn(i, who); } catch (...) { j.failed.store(1, std::memory_order_release); }      // (an exception must not leave a worker thread: reported by the caller)
        j.done.fetch_add(1, std::memory_order_acq_rel); } }
    void worker(int w) {
        long long seen = 0;
        for (;;) {
            long long g = gen.load(std::memory_order_acquire);
            if (g == seen) {
                while ((g = gen.load(std::memory_order_acquire)) == seen && sessions.load(std::memory_order_acquire) > 0) cpu_relax();
                if (g == seen) {
                    std::unique_lock<std::mutex> lk(mu);
                    sleepers.fetch_add(1, std::memory_order_seq_cst);
                    cv.wait(lk, [&] { return gen.load(std::memory_order_seq_cst) != seen || sessions.load(std::memory_order_seq_cst) > 0; });
                    sleepers.fetch_sub(1, std::memory_order_seq_cst);
                    g = gen.load(std::memory_order_acquire);
                    if (g == seen) continue;   // (woken for a session: back to the polling loop)
                }
            }
            seen = g;
            Job &j = slot[g & 1];
            state[w].v.store(2 * g + 1, std::memory_order_seq_cst);
            if (j.tag.load(std::memory_order_seq_cst) == g) drain(j, w + 1);
            state[w].v.store(2 * g, std::memory_order_seq_cst);
        }
    }
public:
    explicit PlanPool(int workers) : state(new WState[std::max(1, workers)]), nworkers(workers) {
        for (int i = 0; i < workers; i++) std::thread([this, i] { worker(i); }).detach();
    }
    int workers() const { return nworkers; }
    // f(i, who) for i < cnt; who = 0 for the calling thread, w + 1 for worker w (at most one item at a time per `who`)
    void run(int cnt, const std::function<void(int, int)> &f) {
        if (cnt <= 0) return;
        if (cnt == 1 || nworkers == 0) { for (int i = 0; i < cnt; i++) f(i, 0); return; }
        std::lock_guard<std::mutex> only(one_caller);
        const long long g = gen.load(std::memory_order_acquire) + 1;
        Job &j = slot[g & 1];
        j.tag.store(-1, std::memory_order_seq_cst);                              // closed: nobody new gets in ...
        for (int w = 0; w < nworkers; w++) while (state[w].v.load(std::memory_order_seq_cst) == 2 * (g - 2) + 1) cpu_relax();     // ... and its last users are out
        j.fn = f; j.n = cnt; j.next.store(0, std::memory_order_relaxed); j.done.store(0, std::memory_order_relaxed); j.failed.store(0, std::memory_order_relaxed);
        j.tag.store(g, std::memory_order_seq_cst);
        gen.store(g, std::memory_order_seq_cst);
        if (sleepers.load(std::memory_order_seq_cst) > 0) { std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
        drain(j, 0);
        while (j.done.load(std::memory_order_acquire) < cnt) cpu_relax();
        j.fn = nullptr;                        // (a worker arriving now finds next >= n and never calls it; fn is only read for an index < n)
        if (j.failed.load(std::memory_order_acquire)) fail(ERR_OOM, "a planner task failed (out of host memory?)");
    }
    // a plan in progress: wakes the workers once (without waiting for them) and keeps them polling until it is over
    void begin_session() {
        sessions.fetch_add(1, std::memory_order_seq_cst);
        if (sleepers.load(std::memory_order_seq_cst) > 0) { std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
    }
    void end_session() { sessions.fetch_sub(1, std::memory_order_seq_cst); }
    struct Session {
        PlanPool *p;
        explicit Session(PlanPool *pool) : p(pool) { if (p) p->begin_session(); }
        ~Session() { if (p) p->end_session(); }
    };
    static PlanPool &get() {
        static PlanPool *pool = [] {
            unsigned hw = std::thread::hardware_concurrency();
            int nt = std::max(1, std::min(16, hw ? (int)hw : 1));
            if (const char *e = getenv("APRILSAM_AMD_PLAN_THREADS")) nt = std::max(1, atoi(e));
            return new PlanPool(nt - 1);       // never destroyed: its threads are parked for the life of the process
        }();
        return *pool;
    }
};

// labels of the regions a worker has in hand (vertices outside them keep label 0) ...
struct Shared {
    std::vector<int> label; int next_label = 0;
    explicit Shared(int N) : label(N, 0) {}
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
    std::vector<int> dist_, loc_; std::vector<char> side_;      // ... and the per-vertex scratch of this evaluator
    int *dist, *loc; char *side;
    Dinic dinic;                // scratch of refine_band
    HopcroftKarp hk_;           // scratch of cut_to_separator (buffers reused across the hundreds of calls of one plan)
    std::vector<int> B0_, B1_, band_, order_; std::vector<char> inL_, inR_; std::vector<std::pair<double, int>> pr_;

    Dissector(int N_, const std::vector<int> &ap_, const std::vector<int> &ai_, const double *xy_, int leaf_, NDTree &t, Shared &sh_)
        : N(N_), ap(ap_), ai(ai_), xy(xy_), leaf(leaf_), tree(t), sh(sh_), dist_(N_, -1), loc_(N_, -1), side_(N_, 0),
          dist(dist_.data()), loc(loc_.data()), side(side_.data()) {}
    // Regions of at least par_min vertices (the top of the tree, which one thread walks alone) have their candidate splits
    // and their refinements evaluated side by side: helper evaluators share this dissector's labels (read only while they
    // run) and own their scratch; every candidate is computed by the same code on the same inputs as in the serial order,
    // so the tree does not depend on the number of threads.
    size_t par_min = 0; PlanPool *pool = nullptr;
    std::vector<std::unique_ptr<Dissector>> helpers;
    Dissector &helper(int k) {                           // (created by the calling thread, before the job that uses them starts)
        if (k == 0) return *this;
        if ((int)helpers.size() <= k) helpers.resize(k + 1);
        for (int q = 1; q <= k; q++) if (!helpers[q]) helpers[q].reset(new Dissector(N, ap, ai, xy, leaf, tree, sh));
        return *helpers[k];
    }
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

    // The component just found by bfs() (visit order in `order`, already labelled Lc) in ASCENDING vertex order, as every split is
    // computed on it.  Regions are kept ascending (the root is 0 .. N-1, parts are filtered from ascending components), so a connected
    // region -- nearly every one -- is its own component, and a large part of a region is a filter pass over it: no sort
    // (std::sort of the root's 3 500 vertices was 0.1 ms of the 0.5 ms its level takes, on the calling thread).
    std::vector<int> ascending_component(const std::vector<int> &verts, bool verts_ascending, const std::vector<int> &order, int Lc) {
        if (verts_ascending && order.size() == verts.size()) return verts;
        std::vector<int> comp;
        if (verts_ascending && order.size() * 8 > verts.size()) { comp.reserve(order.size()); for (int v : verts) if (lab(v) == Lc) comp.push_back(v); }
        else { comp = order; std::sort(comp.begin(), comp.end()); }
        return comp;
    }

    struct alignas(128) Split { std::vector<int> S, P0, P1; double cost = 1e300; bool ok = false; };      // (own cache lines: candidates are filled side by side)
    Split cand_[8], ref_out_;   // candidates of the region being split / result of a refinement (buffers reused from region to region)

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
        Split &out = ref_out_; out.S.clear(); out.P0.clear(); out.P1.clear(); out.ok = false; out.cost = 1e300;
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
        if (out.ok && out.cost < sp.cost) { sp.S.swap(out.S); sp.P0.swap(out.P0); sp.P1.swap(out.P1); sp.cost = out.cost; sp.ok = true; }
    }

    void split_bfs(const std::vector<int> &comp, int L, Split &out) {
        std::vector<int> &order = order_;
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
            const bool asc = std::is_sorted(item.verts.begin(), item.verts.end());
            // connected components of the region
            for (int s : item.verts) {
                if (lab(s) != L) continue;
                bfs(s, L, order);
                int Lc = new_label();
                for (int v : order) { dist[v] = -1; set_lab(v, Lc); }
                std::vector<int> comp = ascending_component(item.verts, asc, order, Lc);
                handle(comp, Lc, item.parent, stack);
            }
        }
    }

    // the split of one connected region of more than `leaf` vertices (null: none found, the region stays a dense node).  A pure
    // function of the region and the graph: it does not touch the tree, and leaves this evaluator's scratch as it found it.
    // ---- the split of one connected region, in pieces (so that the pieces of several regions can run side by side) ----------
    struct CandJob { int slot; double rot; };
    static int refine_min() { static const int v = getenv("APRILSAM_AMD_ND_REF") ? atoi(getenv("APRILSAM_AMD_ND_REF")) : 4; return v; }
    // candidates: slot 3 = the BFS level structure (the longest job: first), the others geometric bisections
    int candidate_jobs(size_t n, CandJob (&jobs)[8]) const {
        static const int T_DIRS = getenv("APRILSAM_AMD_ND_DIRS") ? atoi(getenv("APRILSAM_AMD_ND_DIRS")) : 8;
        // region size (x leaf) above which 4 more directions are tried.  Round 4: 256 (regions of more than 4 096 poses) instead of never --
        // the principal axis of a square region is whatever the noise makes it: on the 1 000 x 1 000 lattice it came out 22 degrees off
        // the axes, none of {0, 45, 90} degrees from it was a lattice direction, and the root separator was a DIAGONAL (1 397 poses
        // instead of 1 000; every level below likewise): with eight directions the top of the tree is what a lattice deserves --
        // sum c_j^2 811 -> 535 GFLOP at 1 M poses, 18.7 -> 17.0 at 100 k; M3500 (3 500 poses) keeps its plan
        static const int T_MORE = getenv("APRILSAM_AMD_ND_MORE") ? atoi(getenv("APRILSAM_AMD_ND_MORE")) : 256;
        int nj = 0;
        jobs[nj++] = { 3, 0.0 };
        jobs[nj++] = { 0, 0.0 };                                         // principal axis
        if ((int)n > T_DIRS * leaf) {                                    // the extra directions only pay near the top of the tree
            jobs[nj++] = { 1, 1.5707963267948966 };                      // orthogonal axis
            jobs[nj++] = { 2, 0.7853981633974483 };                      // diagonal
        }
        if (T_MORE > 0 && (int)n > T_MORE * leaf) {
            jobs[nj++] = { 4, 2.356194490192345 };                       // other diagonal
            jobs[nj++] = { 5, 0.39269908169872414 }; jobs[nj++] = { 6, 1.1780972450961724 }; jobs[nj++] = { 7, 1.9634954084936207 };
        }
        return nj;
    }
    // (on THIS evaluator's scratch; the labels may be another evaluator's, shared)
    void eval_candidate(const std::vector<int> &comp, int L, const CandJob &jb, Split (&cand)[8]) {
        if (jb.slot == 3) split_bfs(comp, L, cand[3]); else split_geometric(comp, L, cand[jb.slot], jb.rot);
    }
    static void choose(Split (&cand)[8], Split *&best, Split *&second) {
        best = nullptr; second = nullptr;
        for (Split &c : cand) {
            if (!c.ok) continue;
            if (!best || c.cost < best->cost) { second = best; best = &c; }
            else if (!second || c.cost < second->cost) second = &c;
        }
    }
    void refine_split(const std::vector<int> &comp, int L, Split *c) {
        static const int T_BAND = getenv("APRILSAM_AMD_ND_BAND") ? atoi(getenv("APRILSAM_AMD_ND_BAND")) : 2;
        for (int pass = 0; pass < 2; pass++) { double before = c->cost; refine_band(comp, L, *c, T_BAND); if (c->cost >= before) break; }
    }
    // the split of one connected region of more than `leaf` vertices (null: none found, the region stays a dense node).  A pure
    // function of the region and the graph: it does not touch the tree, and leaves this evaluator's scratch as it found it.
    Split *best_split(const std::vector<int> &comp, int L) {
        Split (&cand)[8] = cand_;
        for (Split &c : cand) { c.ok = false; c.cost = 1e300; }
        CandJob jobs[8]; const int nj = candidate_jobs(comp.size(), jobs);
        const bool par = pool && par_min > 0 && comp.size() >= par_min;
        if (par) { helper(nj - 1); pool->run(nj, [&](int k, int) { helper(k).eval_candidate(comp, L, jobs[k], cand); }); }
        else for (int k = 0; k < nj; k++) eval_candidate(comp, L, jobs[k], cand);
        Split *best, *second;
        choose(cand, best, second);
        if (best && (int)comp.size() > refine_min() * leaf) {
            if (par && second) {
                // the runner-up is refined at the same time; the rule "only if within 25 % of the REFINED best" is applied afterwards
                // (a runner-up outside it cannot win refined or not, so its refinement is simply ignored: same result as in sequence)
                const double second_cost0 = second->cost;
                helper(1);
                pool->run(2, [&](int k, int) { helper(k).refine_split(comp, L, k == 0 ? best : second); });
                if (second_cost0 > 1.25 * best->cost) second = nullptr;
            } else {
                for (Split *c : { best, second }) {
                    if (!c || (c == second && second->cost > 1.25 * best->cost)) continue;
                    refine_split(comp, L, c);
                }
            }
            if (second && second->cost < best->cost) best = second;
        }
        return best;
    }
    template <class Stack>
    void handle(std::vector<int> &comp, int L, int parent, Stack &stack) {
        Split *best = (int)comp.size() <= leaf ? nullptr : best_split(comp, L);
        if (!best) { for (int v : comp) set_lab(v, -1); new_node(std::move(comp), parent); return; }   // leaf / dense region
        for (int v : best->S) set_lab(v, -1);
        int t = new_node(std::move(best->S), parent);
        stack.push_back({ std::move(best->P0), t });
        stack.push_back({ std::move(best->P1), t });
    }

    // One region of the top of the tree, without touching the tree: its connected components in the order run() meets them,
    // each either a leaf / dense node or a split into separator + two parts.  nested_dissection computes these level by level
    // (regions of a level side by side) and creates the tree nodes afterwards, in the order run() would have.
    struct CompResult { bool leaf = true; std::vector<int> verts, P0, P1; int c0 = -1, c1 = -1; };      // verts: the leaf, or the separator; c0 / c1: results of the parts (-1: set aside)
    struct RegionResult { std::deque<CompResult> comps; };       // (a deque: results are filled in through pointers while more are appended)
    void process_region(const std::vector<int> &verts, RegionResult &out) {
        std::vector<int> &order = comp_order_;
        const int L = new_label();
        for (int v : verts) set_lab(v, L);
        const bool asc = std::is_sorted(verts.begin(), verts.end());
        for (int s : verts) {
            if (lab(s) != L) continue;
            bfs(s, L, order);
            const int Lc = new_label();
            for (int v : order) { dist[v] = -1; set_lab(v, Lc); }
            std::vector<int> comp = ascending_component(verts, asc, order, Lc);
            out.comps.emplace_back();
            CompResult &cr = out.comps.back();
            Split *best = (int)comp.size() <= leaf ? nullptr : best_split(comp, Lc);
            if (!best) { cr.leaf = true; cr.verts = std::move(comp); }
            else { cr.leaf = false; cr.verts = std::move(best->S); cr.P0 = std::move(best->P0); cr.P1 = std::move(best->P1); }
        }
    }
    std::vector<int> comp_order_;

    // The same for ALL regions of a level that has only a few (the first levels of the tree), with the pieces of their splits
    // side by side: every candidate of every connected component in one job, every refinement in another.  Components and
    // labels are this evaluator's; the candidates run on per-thread helpers that share its labels.
    struct CompWork { std::vector<int> comp; int L = 0; CompResult *out = nullptr; Split cand[8]; Split *best = nullptr, *second = nullptr; double second_cost0 = 0; };
    void process_regions(const std::vector<const std::vector<int> *> &regions, const std::vector<RegionResult *> &outs) {
        static const bool prof = getenv("APRILSAM_AMD_PLAN_PROFILE") != nullptr;
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double tp0 = prof ? now() : 0;
        std::vector<std::unique_ptr<CompWork>> work;
        std::vector<int> &order = comp_order_;
        for (size_t r = 0; r < regions.size(); r++) {
            const std::vector<int> &verts = *regions[r];
            const int L = new_label();
            for (int v : verts) set_lab(v, L);
            const bool asc = std::is_sorted(verts.begin(), verts.end());
            for (int s : verts) {
                if (lab(s) != L) continue;
                bfs(s, L, order);
                const int Lc = new_label();
                for (int v : order) { dist[v] = -1; set_lab(v, Lc); }
                std::vector<int> comp = ascending_component(verts, asc, order, Lc);
                outs[r]->comps.emplace_back();
                CompResult &cr = outs[r]->comps.back();
                if ((int)comp.size() <= leaf) { cr.leaf = true; cr.verts = std::move(comp); continue; }
                work.emplace_back(new CompWork());
                work.back()->comp = std::move(comp); work.back()->L = Lc; work.back()->out = &cr;
            }
        }
        if (helpers.size() < (size_t)pool->workers() + 1) helpers.resize((size_t)pool->workers() + 1);
        auto mine = [&](int who) -> Dissector & {            // evaluator of the executing thread (who = 0: this one)
            if (who == 0) return *this;
            if (!helpers[who]) helpers[who].reset(new Dissector(N, ap, ai, xy, leaf, tree, sh));
            return *helpers[who];
        };
        struct J { CompWork *w; CandJob jb; };
        std::vector<J> jobs;
        for (auto &w : work) { CandJob cj[8]; const int nj = candidate_jobs(w->comp.size(), cj); for (int k = 0; k < nj; k++) jobs.push_back({ w.get(), cj[k] }); }
        std::stable_sort(jobs.begin(), jobs.end(), [](const J &a, const J &b) { return (a.jb.slot == 3) > (b.jb.slot == 3); });      // the long ones first
        const double tp1 = prof ? now() : 0;
        pool->run((int)jobs.size(), [&](int k, int who) { mine(who).eval_candidate(jobs[k].w->comp, jobs[k].w->L, jobs[k].jb, jobs[k].w->cand); });
        const double tp2 = prof ? now() : 0;
        struct R { CompWork *w; Split *c; };
        std::vector<R> refs;
        for (auto &w : work) {
            choose(w->cand, w->best, w->second);
            if (w->best && (int)w->comp.size() > refine_min() * leaf) {
                refs.push_back({ w.get(), w->best });
                if (w->second) { w->second_cost0 = w->second->cost; refs.push_back({ w.get(), w->second }); }
            }
        }
        pool->run((int)refs.size(), [&](int k, int who) { mine(who).refine_split(refs[k].w->comp, refs[k].w->L, refs[k].c); });
        const double tp3 = prof ? now() : 0;
        struct Fin { bool on; double a, b, c, d; size_t n; std::function<double()> now_; ~Fin() { if (on) fprintf(stderr, "aprilsam_amd dissection: %zu region(s): components %.3f candidates %.3f refinement %.3f results %.3f ms\n", n, b - a, c - b, d - c, now_() - d); } } fin_{ prof, tp0, tp1, tp2, tp3, regions.size(), now };
        for (auto &w : work) {
            Split *best = w->best, *second = w->second;
            if (best && (int)w->comp.size() > refine_min() * leaf) {
                if (second && w->second_cost0 > 1.25 * best->cost) second = nullptr;      // (as in best_split: the speculative refinement is ignored)
                if (second && second->cost < best->cost) best = second;
            }
            CompResult &cr = *w->out;
            if (!best) { cr.leaf = true; cr.verts = std::move(w->comp); }
            else { cr.leaf = false; cr.verts = std::move(best->S); cr.P0 = std::move(best->P0); cr.P1 = std::move(best->P1); }
        }
    }
};

}  // namespace

static PlanPool *plan_pool_or_null() {
    if (const char *e = getenv("APRILSAM_AMD_PLAN_THREADS")) if (atoi(e) <= 1) return nullptr;      // 1 = single-threaded planning
    PlanPool &p = PlanPool::get();
    return p.workers() > 0 ? &p : nullptr;
}
static thread_local int g_plan_sessions = 0;      // sessions opened by this thread (plan_parallel_for only fans out inside one)
PlanSession::PlanSession(int n_nodes, int min_nodes) {
    if (n_nodes < min_nodes) return;
    PlanPool *p = plan_pool_or_null();
    if (!p) return;
    on = true; g_plan_sessions++;
    p->begin_session();
}
PlanSession::~PlanSession() { if (on) { g_plan_sessions--; PlanPool::get().end_session(); } }
void plan_parallel_for(int n, int grain, const std::function<void(int, int)> &body) {
    if (n <= 0) return;
    PlanPool *p = g_plan_sessions > 0 ? plan_pool_or_null() : nullptr;
    const int chunks = p ? std::min((p->workers() + 1) * 4, std::max(1, n / std::max(1, grain))) : 1;
    if (chunks <= 1) { body(0, n); return; }
    p->run(chunks, [&](int i, int) { body((int)((long long)n * i / chunks), (int)((long long)n * (i + 1) / chunks)); });
}

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
    int nthreads = 0;                                                     // 0: as many as the pool has
    if (const char *e = getenv("APRILSAM_AMD_PLAN_THREADS")) nthreads = std::max(1, atoi(e));      // 1 = single-threaded planning
    PlanPool *pool = (nthreads == 1 || N < 1024) ? nullptr : &PlanPool::get();
    if (pool && pool->workers() == 0) pool = nullptr;
    PlanPool::Session session(pool);
    const bool prof = getenv("APRILSAM_AMD_PLAN_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = prof ? now() : 0;
    if (!pool) {
        Dissector d(N, adj_ptr, adj, xy, leaf_nodes, tree, sh);
        d.run(std::move(top), defer_below, defer_below ? &tasks : nullptr);
    } else {
        // The same top of the tree, level by level: the regions of one level are independent -- few of them (the first levels):
        // one after the other, each with its candidate splits side by side; several: side by side, one evaluator per thread.
        // The tree nodes are created afterwards by replaying run()'s order over the results, so the numbering is the serial one.
        using RR = Dissector::RegionResult;
        std::vector<std::unique_ptr<RR>> results;
        struct Pending { std::vector<int> verts; int res; };
        std::vector<Pending> level, next;
        results.emplace_back(new RR()); level.push_back({ std::move(top[0].verts), 0 });
        const int nev = pool->workers() + 1;
        std::vector<std::unique_ptr<Shared>> ev_sh(nev); std::vector<std::unique_ptr<Dissector>> ev(nev);
        auto evaluator = [&](int who) -> Dissector & {            // (created by the thread that first needs it: its pages land near that thread)
            if (!ev[who]) { ev_sh[who].reset(new Shared(N)); ev[who].reset(new Dissector(N, adj_ptr, adj, xy, leaf_nodes, tree, *ev_sh[who])); }
            return *ev[who];
        };
        evaluator(0).pool = pool; evaluator(0).par_min = 1;
        std::string lvl_times;
        while (!level.empty()) {
            const double tl0 = prof ? now() : 0;
            struct LevelStamp { std::string &out; double t0; size_t n; bool on; std::function<double()> now_; ~LevelStamp() { if (on) { char b[64]; snprintf(b, sizeof b, " %zu:%.3f", n, now_() - t0); out += b; } } } stamp_{ lvl_times, tl0, level.size(), prof, now };
            if (level.size() < 3) {
                std::vector<const std::vector<int> *> rv; std::vector<RR *> ro;
                for (Pending &pd : level) { rv.push_back(&pd.verts); ro.push_back(results[pd.res].get()); }
                evaluator(0).process_regions(rv, ro);
            }
            else {
                evaluator(0).pool = nullptr;            // (no job inside a job: the calling thread takes regions like everybody else)
                pool->run((int)level.size(), [&](int i, int who) { evaluator(who).process_region(level[i].verts, *results[level[i].res]); });
                evaluator(0).pool = pool;
            }
            next.clear();
            for (Pending &pd : level)
                for (Dissector::CompResult &cr : results[pd.res]->comps) {
                    if (cr.leaf) continue;
                    if (cr.P0.size() > defer_below) { cr.c0 = (int)results.size(); results.emplace_back(new RR()); next.push_back({ std::move(cr.P0), cr.c0 }); }
                    if (cr.P1.size() > defer_below) { cr.c1 = (int)results.size(); results.emplace_back(new RR()); next.push_back({ std::move(cr.P1), cr.c1 }); }
                }
            level.swap(next);
        }
        if (prof) fprintf(stderr, "aprilsam_amd dissection N=%d, levels of the top of the tree (regions:ms):%s\n", N, lvl_times.c_str());
        // replay: run()'s stack discipline over the results
        struct Ent { int res; std::vector<int> verts; int parent; };
        std::vector<Ent> stack; stack.push_back({ 0, {}, -1 });
        auto new_node = [&](std::vector<int> &&verts, int parent) {
            const int id = (int)tree.nodes.size();
            tree.nodes.emplace_back(); tree.nodes.back().verts = std::move(verts);
            if (parent < 0) tree.roots.push_back(id); else tree.nodes[parent].children.push_back(id);
            return id;
        };
        while (!stack.empty()) {
            Ent e = std::move(stack.back()); stack.pop_back();
            if (e.res < 0) { tasks.push_back({ std::move(e.verts), e.parent }); continue; }
            for (Dissector::CompResult &cr : results[e.res]->comps) {
                if (cr.leaf) { new_node(std::move(cr.verts), e.parent); continue; }
                const int t = new_node(std::move(cr.verts), e.parent);
                stack.push_back({ cr.c0, std::move(cr.P0), t });
                stack.push_back({ cr.c1, std::move(cr.P1), t });
            }
        }
    }
    const double t1 = prof ? now() : 0;
    if (tasks.empty()) return;
    const int nt = (int)tasks.size();
    std::vector<NDTree> sub(nt);
    auto work = [&](int i) {
        std::vector<Item> st; Item it; it.verts = std::move(tasks[i].verts); it.parent = -1; st.push_back(std::move(it));
        Shared mine(N);           // private labels and scratch: sharing one set of arrays between threads is correct (disjoint
        Dissector d(N, adj_ptr, adj, xy, leaf_nodes, sub[i], mine);      // vertices) but falsely shares cache lines -- measured: no speed-up
        d.run(std::move(st));
    };
    if (!pool) { for (int i = 0; i < nt; i++) work(i); }
    else pool->run(nt, [&](int i, int) { work(i); });
    if (prof) fprintf(stderr, "aprilsam_amd dissection N=%d: top of the tree %.3f ms, %d subtrees %.3f ms (%d worker threads)\n", N, t1 - t0, nt, now() - t1, pool ? pool->workers() : 0);
    for (int i = 0; i < nt; i++) {
        const int off = (int)tree.nodes.size(), parent = tasks[i].parent;
        for (NDTree::Node &nd : sub[i].nodes) { for (int &c : nd.children) c += off; tree.nodes.push_back(std::move(nd)); }
        for (int r : sub[i].roots) { if (parent < 0) tree.roots.push_back(r + off); else tree.nodes[parent].children.push_back(r + off); }
    }
}

}  // namespace asam

// solver.hip.cpp — host runtime of the MI355X Gauss-Newton step: packs the caller's graph objects
// (reference ABI, include/aprilsam_amd.h PART 1) into SoA arrays, keeps ordering / symbolic plan /
// HBM-resident fronts in a side context keyed by the param pointer, and drives the HIP kernels on one
// stream per context.  Entry points (C ABI) are at the bottom.
//
// Reference call stack this replaces: aprilsam.c:87-375 (april_graph_cholesky) — see SURVEY.md §3.1.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/aprilsam_amd.h"
#include "kernels.hip.h"
#include "plan.h"
#include "refmodel.h"
#include "solver.h"

namespace asam {

// ------------------------------------------------------------------------------------------------------
// utilities
// ------------------------------------------------------------------------------------------------------
[[noreturn]] static void fatal(const char *msg) {
    fprintf(stderr, "aprilsam_amd: FATAL: %s\n", msg);
    fflush(stderr);
    abort();
}
#define HIPCHECK(expr)                                                                                 \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "aprilsam_amd: FATAL: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            fflush(stderr);                                                                            \
            abort();                                                                                   \
        }                                                                                              \
    } while (0)

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

Options g_opt;
static std::once_flag g_opt_once;
static void load_env_options() {
    std::call_once(g_opt_once, [] {
        auto envd = [](const char *n, double *v) { const char *s = getenv(n); if (s && *s) *v = atof(s); };
        double v;
        v = g_opt.leaf_nodes; envd("APRILSAM_AMD_LEAF_NODES", &v); g_opt.leaf_nodes = (int)v;
        v = g_opt.deterministic; envd("APRILSAM_AMD_DETERMINISTIC", &v); g_opt.deterministic = (int)v;
        v = g_opt.use_graph; envd("APRILSAM_AMD_USE_GRAPH", &v); g_opt.use_graph = (int)v;
        v = g_opt.device_timing; envd("APRILSAM_AMD_DEVICE_TIMING", &v); g_opt.device_timing = (int)v;
        v = g_opt.trust_factor_cache; envd("APRILSAM_AMD_TRUST_FACTOR_CACHE", &v); g_opt.trust_factor_cache = (int)v;
        v = g_opt.small_lds_kb; envd("APRILSAM_AMD_SMALL_LDS_KB", &v); g_opt.small_lds_kb = (int)v;
        v = g_opt.medium_lds_kb; envd("APRILSAM_AMD_MEDIUM_LDS_KB", &v); g_opt.medium_lds_kb = (int)v;
    });
}

static int g_device = -1;
static int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
static void ensure_device() {
    static std::once_flag once;
    std::call_once(once, [] {
        load_env_options();
        int n = device_count();
        if (n <= 0) fatal("no HIP device visible: the april_graph_cholesky* / april_graph_chi2 entry points of "
                          "libaprilsam_amd.so run on an AMD GPU only (there is no CPU fallback)");
        if (g_device < 0) {
            const char *lr = getenv("LOCAL_RANK");
            g_device = lr ? atoi(lr) % n : 0;
        }
    });
    HIPCHECK(hipSetDevice(g_device));
}

// grow-only device / pinned-host buffers
template <class T> struct DBuf {
    T *p = nullptr; size_t cap = 0;
    void need(size_t n) {
        if (n <= cap) return;
        if (p) HIPCHECK(hipFree(p));
        size_t c = std::max(n, cap + cap / 2);
        HIPCHECK(hipMalloc((void **)&p, c * sizeof(T)));
        cap = c;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
template <class T> struct HBuf {
    T *p = nullptr; size_t cap = 0;
    void need(size_t n, bool keep = false) {
        if (n <= cap) return;
        size_t c = std::max(n, cap + cap / 2);
        T *q = nullptr;
        HIPCHECK(hipHostMalloc((void **)&q, c * sizeof(T), hipHostMallocDefault));
        if (p) { if (keep) memcpy(q, p, cap * sizeof(T)); HIPCHECK(hipHostFree(p)); }
        p = q; cap = c;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// ------------------------------------------------------------------------------------------------------
// packed graph (SoA, host pinned + device) — one per april_graph_t pointer
// ------------------------------------------------------------------------------------------------------
struct GraphPack {
    int N = 0, F = 0;                  // packed counts
    std::vector<const void *> fptr;    // factor object pointers already packed (cache validation)
    HBuf<int> h_fa, h_fb;
    HBuf<double> h_z, h_W, h_state, h_lp, h_dx;
    DBuf<int> d_fa, d_fb;
    DBuf<double> d_z, d_W, d_state, d_lp, d_dx, d_chi2f, d_scalar;
    int F_on_device = 0;               // factors already uploaded
    int F_cap = 0;                     // device capacity (factors) of d_fa/d_fb/d_z/d_W/d_chi2f
    hipStream_t stream = nullptr;
    HBuf<double> h_scalar;
    void release() {
        h_fa.release(); h_fb.release(); h_z.release(); h_W.release(); h_state.release(); h_lp.release(); h_dx.release();
        d_fa.release(); d_fb.release(); d_z.release(); d_W.release(); d_state.release(); d_lp.release(); d_dx.release();
        d_chi2f.release(); d_scalar.release(); h_scalar.release();
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
};

static std::mutex g_mu;
static std::unordered_map<const void *, std::unique_ptr<GraphPack>> g_packs;

static GraphPack &pack_for(const april_graph_t *g) {
    auto it = g_packs.find(g);
    if (it == g_packs.end()) {
        auto p = std::make_unique<GraphPack>();
        HIPCHECK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        it = g_packs.emplace(g, std::move(p)).first;
    }
    return *it->second;
}
void drop_graph_pack(const april_graph_t *g) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_packs.find(g);
    if (it != g_packs.end()) { it->second->release(); g_packs.erase(it); }
}

static inline int zsize(const zarray_t *z) { return z ? z->size : 0; }

// (re)pack factors [from, F): ids, z, W.  Unknown factor types are fatal (no host-fallback vtable yet).
static void pack_factors(GraphPack &gp, const april_graph_t *g) {
    const int F = zsize(g->factors);
    april_graph_factor_t **fs = (april_graph_factor_t **)g->factors->data;
    int from = gp.F;
    bool valid = g_opt.trust_factor_cache && from <= F && (int)gp.fptr.size() == from &&
                 (from == 0 || memcmp(gp.fptr.data(), fs, sizeof(void *) * from) == 0);
    if (!valid) { from = 0; gp.F_on_device = 0; }
    gp.h_fa.need(F, true); gp.h_fb.need(F, true); gp.h_z.need((size_t)3 * F, true); gp.h_W.need((size_t)9 * F, true);
    gp.fptr.resize(F);
    const int N = zsize(g->nodes);
    for (int i = from; i < F; i++) {
        const april_graph_factor_t *f = fs[i];
        gp.fptr[i] = f;
        int a = -1, b = -1;
        if (f->type == APRIL_GRAPH_FACTOR_XYT_TYPE && f->nnodes == 2) { a = f->nodes[0]; b = f->nodes[1]; }
        else if (f->type == APRIL_GRAPH_FACTOR_XYTPOS_TYPE && f->nnodes == 1) { a = f->nodes[0]; b = -1; }
        else {
            fprintf(stderr, "aprilsam_amd: FATAL: factor %d has type %d / %d nodes; only xyt (1) and xytpos (2) factors are "
                            "evaluated on the device\n", i, f->type, f->nnodes);
            abort();
        }
        if (a < 0 || a >= N || b >= N || a == b) { fprintf(stderr, "aprilsam_amd: FATAL: factor %d references node out of range\n", i); abort(); }
        gp.h_fa.p[i] = a; gp.h_fb.p[i] = b;
        memcpy(gp.h_z.p + (size_t)3 * i, f->u.common.z, 24);
        memcpy(gp.h_W.p + (size_t)9 * i, f->u.common.W->data, 72);
    }
    gp.F = F;
}
static void upload_factors(GraphPack &gp) {
    const int F = gp.F;
    if (F > gp.F_cap) {           // reallocation loses the old content: re-upload everything
        gp.F_cap = std::max(F, gp.F_cap + gp.F_cap / 2 + 64);
        gp.d_fa.need(gp.F_cap); gp.d_fb.need(gp.F_cap); gp.d_z.need((size_t)3 * gp.F_cap); gp.d_W.need((size_t)9 * gp.F_cap);
        gp.d_chi2f.need(gp.F_cap);
        gp.F_on_device = 0;
    }
    const int f0 = gp.F_on_device;
    if (F > f0) {
        size_t n = F - f0;
        HIPCHECK(hipMemcpyAsync(gp.d_fa.p + f0, gp.h_fa.p + f0, n * 4, hipMemcpyHostToDevice, gp.stream));
        HIPCHECK(hipMemcpyAsync(gp.d_fb.p + f0, gp.h_fb.p + f0, n * 4, hipMemcpyHostToDevice, gp.stream));
        HIPCHECK(hipMemcpyAsync(gp.d_z.p + (size_t)3 * f0, gp.h_z.p + (size_t)3 * f0, n * 24, hipMemcpyHostToDevice, gp.stream));
        HIPCHECK(hipMemcpyAsync(gp.d_W.p + (size_t)9 * f0, gp.h_W.p + (size_t)9 * f0, n * 72, hipMemcpyHostToDevice, gp.stream));
    }
    gp.F_on_device = F;
    gp.d_scalar.need(8); gp.h_scalar.need(8);
}
// states (and l_points) of all nodes -> pinned host -> device
static void pack_states(GraphPack &gp, const april_graph_t *g, bool with_lp) {
    const int N = zsize(g->nodes);
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    gp.h_state.need((size_t)3 * N); gp.h_lp.need((size_t)3 * N); gp.h_dx.need((size_t)3 * N);
    for (int i = 0; i < N; i++) {
        const april_graph_node_t *n = ns[i];
        if (n->type != APRIL_GRAPH_NODE_XYT_TYPE || n->length != 3) fatal("only xyt nodes (type 100, 3 DoF) are supported (aprilsam.h:94)");
        memcpy(gp.h_state.p + (size_t)3 * i, n->state, 24);
        if (with_lp) memcpy(gp.h_lp.p + (size_t)3 * i, n->l_point, 24);
    }
    gp.N = N;
    gp.d_state.need((size_t)3 * N); gp.d_lp.need((size_t)3 * N); gp.d_dx.need((size_t)3 * N);
    HIPCHECK(hipMemcpyAsync(gp.d_state.p, gp.h_state.p, (size_t)24 * N, hipMemcpyHostToDevice, gp.stream));
    if (with_lp) HIPCHECK(hipMemcpyAsync(gp.d_lp.p, gp.h_lp.p, (size_t)24 * N, hipMemcpyHostToDevice, gp.stream));
}

// ------------------------------------------------------------------------------------------------------
// solver context — one per april_graph_cholesky_param_t pointer
// ------------------------------------------------------------------------------------------------------
enum { K_LINEARIZE = 0, K_FRONT_SMALL, K_FRONT_MEDIUM, K_ASSEMBLE_BIG, K_PANEL_BIG, K_SYRK_BIG, K_BACKSOLVE, K_UPDATE, NKERN };
static const char *const KNAMES[NKERN] = { "k_linearize", "k_front_small", "k_front_medium", "k_assemble_big", "k_panel_big", "k_syrk_big", "k_backsolve", "k_update_states" };
struct Launch { int list_off, pre_off, n, grid; };     // offsets into the int launch-table buffer

struct LevelPlan {
    int small_off = 0, n_small = 0; size_t small_lds = 0;      // fronts handled by k_front_small
    int med_off = 0, n_med = 0; size_t med_lds = 0;            // fronts handled by k_front_medium
    int n_big = 0; size_t asm_lds = 0;
    Launch asm_big{};                                          // k_assemble_big
    std::vector<Launch> panel, syrk;                           // per panel step
    int all_off = 0, n_all = 0; size_t solve_lds = 0;          // every front (k_backsolve)
};

struct Context {
    Plan plan;
    bool have_plan = false;
    std::vector<int> pat;                 // factor node ids the plan was built for (2 per factor)
    int patN = 0;
    // device copies of the plan
    DBuf<int> d_i32; DBuf<FrontDesc> d_fd; DBuf<DestRec> d_dest; DBuf<ChildRec> d_child; DBuf<double> d_lambda;
    DevPlan dp{};
    DBuf<int> d_tab;                      // launch tables
    std::vector<LevelPlan> levels;
    DBuf<unsigned char> d_swap;
    DBuf<int> d_pos;
    DBuf<long long> d_prof;
    // numeric state
    DBuf<double> d_pool, d_H, d_x;
    DBuf<int> d_bad;
    HBuf<int> h_bad;
    std::vector<double> h_lambda;
    aprilsam_amd_stats_t st{};
    hipEvent_t ev[8] = {};
    bool have_events = false;
    // per-kernel HIP-event timing (instrumented passes only)
    double k_ms[NKERN] = {}; long long k_calls[NKERN] = {};
    std::vector<hipEvent_t> k_ev; std::vector<int> k_ids;
    // incremental bookkeeping (aprilsam.c:741-751, 566-575)
    bool have_fact = false;               // a batch factorisation exists (reference: param->chol != NULL)
    int batch_nodes = 0;                  // #nodes at the last batch step (those carry the Tikhonov term)
    RefModel model;                       // the reference's tree / counters (refmodel.cpp), rebuilt lazily after a batch
    int batch_factors = 0;                // #factors at the last batch step
    // captured numeric phase
    hipGraphExec_t gexec = nullptr;
    const void *gexec_key = nullptr;      // GraphPack the graph was captured against
    void release() {
        d_i32.release(); d_fd.release(); d_dest.release(); d_child.release(); d_lambda.release(); d_tab.release(); d_swap.release(); d_pos.release();
        d_pool.release(); d_H.release(); d_x.release(); d_bad.release(); h_bad.release();
        if (gexec) (void)hipGraphExecDestroy(gexec);
        gexec = nullptr;
        if (have_events) for (auto &e : ev) (void)hipEventDestroy(e);
        have_events = false;
        for (auto &e : k_ev) (void)hipEventDestroy(e);
        k_ev.clear();
    }
};
static std::unordered_map<const void *, std::unique_ptr<Context>> g_ctx;

static Context &ctx_for(const april_graph_cholesky_param_t *p) {
    auto it = g_ctx.find(p);
    if (it == g_ctx.end()) it = g_ctx.emplace(p, std::make_unique<Context>()).first;
    return *it->second;
}
void drop_context(const april_graph_cholesky_param_t *p) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(p);
    if (it != g_ctx.end()) { it->second->release(); g_ctx.erase(it); }
}
bool get_stats(const april_graph_cholesky_param_t *p, aprilsam_amd_stats_t *out) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(p);
    if (it == g_ctx.end()) return false;
    *out = it->second->st;
    return true;
}

// upload the symbolic plan and build the per-level launch tables
static void upload_plan(Context &c, hipStream_t s) {
    const Plan &P = c.plan;
    if (c.gexec) { (void)hipGraphExecDestroy(c.gexec); c.gexec = nullptr; }
    // ---- descriptors + index arrays ----------------------------------------------------------------------------
    std::vector<FrontDesc> fd(P.nF);
    std::vector<ChildRec> ch(std::max<size_t>(1, P.ch_idx.size()));
    for (int t = 0; t < P.nF; t++) {
        FrontDesc &d = fd[t];
        memset(&d, 0, sizeof(d));
        d.off = P.f_off[t]; d.nsb = P.f_nsb[t]; d.nub = P.f_nub[t]; d.first = P.f_first[t];
        d.dest_begin = P.dest_front_ptr[t]; d.dest_end = P.dest_front_ptr[t + 1];
        d.ch_begin = P.ch_ptr[t]; d.ch_end = P.ch_ptr[t + 1];
        d.rows_begin = (int)P.f_rows_ptr[t]; d.parent = P.f_parent[t];
    }
    for (size_t k = 0; k < P.ch_idx.size(); k++) {
        const int cfr = P.ch_idx[k];
        ChildRec &r = ch[k];
        r.cR = P.rows(cfr); r.cnu = P.f_nub[cfr];
        r.uoff = P.f_off[cfr] + (long long)(3 * P.f_nsb[cfr]) * r.cR + 3 * P.f_nsb[cfr];
        r.rel_begin = (int)P.f_rows_ptr[cfr]; r.pad = 0;
    }
    std::vector<int> i32;
    auto put32 = [&](const std::vector<int> &v) { size_t o = i32.size(); i32.insert(i32.end(), v.begin(), v.end()); if (v.empty()) i32.push_back(0); return o; };
    size_t o_rows = put32(P.f_rows), o_rel = put32(P.f_rel), o_sb = put32(P.slot_blk), o_sr = put32(P.slot_rhs);
    c.d_i32.need(i32.size()); c.d_fd.need(fd.size()); c.d_child.need(ch.size()); c.d_dest.need(std::max<size_t>(1, P.dest.size()));
    HIPCHECK(hipMemcpyAsync(c.d_i32.p, i32.data(), i32.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(c.d_fd.p, fd.data(), fd.size() * sizeof(FrontDesc), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(c.d_child.p, ch.data(), ch.size() * sizeof(ChildRec), hipMemcpyHostToDevice, s));
    static_assert(sizeof(DestRec) == sizeof(Plan::DestRec), "DestRec layout");
    if (!P.dest.empty()) HIPCHECK(hipMemcpyAsync(c.d_dest.p, P.dest.data(), P.dest.size() * sizeof(DestRec), hipMemcpyHostToDevice, s));
    c.d_lambda.need(std::max(1, P.N));
    DevPlan &d = c.dp;
    d.nF = P.nF;
    d.fd = c.d_fd.p; d.dest = c.d_dest.p; d.child = c.d_child.p;
    d.f_rows = c.d_i32.p + o_rows; d.f_rel = c.d_i32.p + o_rel; d.slot_blk = c.d_i32.p + o_sb; d.slot_rhs = c.d_i32.p + o_sr;
    d.lambda = c.d_lambda.p;
    d.prof = nullptr;
    if (getenv("APRILSAM_AMD_KPROF")) { c.d_prof.need((size_t)8 * P.nF); HIPCHECK(hipMemsetAsync(c.d_prof.p, 0, (size_t)64 * P.nF, s)); d.prof = c.d_prof.p; }
    c.d_swap.need(std::max(1, P.F)); c.d_pos.need(std::max(1, P.N));
    HIPCHECK(hipMemcpyAsync(c.d_swap.p, P.fac_swap.data(), P.F, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(c.d_pos.p, P.pos.data(), (size_t)P.N * 4, hipMemcpyHostToDevice, s));

    // ---- launch tables -------------------------------------------------------------------------------------
    std::vector<int> tab;
    c.levels.assign(P.nLevels, LevelPlan());
    const size_t small_max = (size_t)g_opt.small_lds_kb * 1024, med_max = (size_t)g_opt.medium_lds_kb * 1024;
    for (int l = 0; l < P.nLevels; l++) {
        LevelPlan &L = c.levels[l];
        std::vector<int> small, med, big;
        size_t maxm = 0;
        for (int k = P.lev_ptr[l]; k < P.lev_ptr[l + 1]; k++) {
            int t = P.lev_fronts[k];
            const int R = P.rows(t), C = P.cols(t);
            maxm = std::max<size_t>(maxm, C);
            const size_t lds_s = small_front_lds(R, C), lds_m = medium_front_lds(R);
            if (lds_s <= small_max) { small.push_back(t); L.small_lds = std::max(L.small_lds, lds_s); }
            else if (lds_m <= med_max) { med.push_back(t); L.med_lds = std::max(L.med_lds, lds_m); }
            else { big.push_back(t); L.asm_lds = std::max(L.asm_lds, scratch_bytes(R)); }
        }
        L.all_off = (int)tab.size(); L.n_all = P.lev_ptr[l + 1] - P.lev_ptr[l];
        tab.insert(tab.end(), P.lev_fronts.begin() + P.lev_ptr[l], P.lev_fronts.begin() + P.lev_ptr[l + 1]);
        L.solve_lds = (maxm + NB + 8) * 8;
        L.small_off = (int)tab.size(); L.n_small = (int)small.size();
        tab.insert(tab.end(), small.begin(), small.end());
        L.med_off = (int)tab.size(); L.n_med = (int)med.size();
        tab.insert(tab.end(), med.begin(), med.end());
        L.n_big = (int)big.size();
        if (big.empty()) continue;
        std::sort(big.begin(), big.end(), [&](int a, int b) { return P.f_nsb[a] != P.f_nsb[b] ? P.f_nsb[a] > P.f_nsb[b] : a < b; });
        int list_off = (int)tab.size();
        tab.insert(tab.end(), big.begin(), big.end());
        auto make = [&](int nact, auto count) {
            Launch La; La.list_off = list_off; La.n = nact; La.pre_off = (int)tab.size();
            int acc = 0; tab.push_back(0);
            for (int i = 0; i < nact; i++) { acc += count(big[i]); tab.push_back(acc); }
            La.grid = acc;
            return La;
        };
        L.asm_big = make((int)big.size(), [&](int t) { return asm_chunks(P.f_nsb[t] + P.f_nub[t]); });
        int steps = (3 * P.f_nsb[big[0]] + NB - 1) / NB;
        for (int sidx = 0; sidx < steps; sidx++) {
            int nact = 0;
            while (nact < (int)big.size() && 3 * P.f_nsb[big[nact]] > sidx * NB) nact++;
            L.panel.push_back(make(nact, [&](int t) { return panel_tiles(P.rows(t), 3 * P.f_nsb[t], sidx); }));
            L.syrk.push_back(make(nact, [&](int t) { return syrk_tiles(P.rows(t), P.cols(t), 3 * P.f_nsb[t], sidx); }));
        }
    }
    if (tab.empty()) tab.push_back(0);
    c.d_tab.need(tab.size());
    HIPCHECK(hipMemcpyAsync(c.d_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHECK(hipStreamSynchronize(s));      // host vectors above go out of scope

    c.d_pool.need((size_t)std::max<int64_t>(P.pool_doubles, 1));
    c.d_H.need((size_t)9 * std::max(1, P.n_slots)); c.d_x.need((size_t)3 * std::max(1, P.N));
    c.d_bad.need(4); c.h_bad.need(4);
    c.st.n_fronts = P.nF; c.st.n_levels = P.nLevels; c.st.max_front_rows = P.max_rows;
    c.st.nnz_L = P.nnzL; c.st.flops_factor = P.flops; c.st.bytes_fronts = 8.0 * (double)P.pool_doubles;
}

static void set_small_attr() {
    static std::once_flag once;
    std::call_once(once, [] {
        HIPCHECK(hipFuncSetAttribute((const void *)k_front_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_front_medium, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_backsolve, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_assemble_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    });
}

// enqueue: linearise -> per level {assemble+factor} -> back substitution -> state update
// ev != null: record stage events (0 start, 1 after linearise, 2 after factor, 3 after solve+update)
// ktime: bracket EVERY kernel launch with its own HIP event pair on this stream (c.k_ev / c.k_ids)
static void enqueue_numeric(Context &c, GraphPack &gp, hipStream_t s, hipEvent_t *ev, bool unary_at_lp = false, bool ktime = false) {
    const Plan &P = c.plan;
    const int F = P.F, N = P.N;
    size_t nev = 0;
    if (ktime) c.k_ids.clear();
    auto tic = [&](int id) {
        if (!ktime) return;
        if (c.k_ev.size() < nev + 2) { c.k_ev.resize(nev + 2); HIPCHECK(hipEventCreate(&c.k_ev[nev])); HIPCHECK(hipEventCreate(&c.k_ev[nev + 1])); }
        HIPCHECK(hipEventRecord(c.k_ev[nev], s));
        c.k_ids.push_back(id);
    };
    auto toc = [&]() { if (ktime) { HIPCHECK(hipEventRecord(c.k_ev[nev + 1], s)); nev += 2; } };
    if (ev) HIPCHECK(hipEventRecord(ev[0], s));
    HIPCHECK(hipMemsetAsync(c.d_bad.p, 0, 4, s));
    if (c.dp.prof) HIPCHECK(hipMemsetAsync(c.d_prof.p, 0, (size_t)64 * P.nF, s));
    tic(K_LINEARIZE);
    hipLaunchKernelGGL(k_linearize, dim3((F + TPB - 1) / TPB), dim3(TPB), 0, s, 0, F, gp.d_fa.p, gp.d_fb.p, gp.d_z.p, gp.d_W.p,
                       gp.d_lp.p, unary_at_lp ? gp.d_lp.p : gp.d_state.p, c.d_swap.p, c.dp.slot_blk, c.dp.slot_rhs, c.d_H.p);
    toc();
    if (ev) HIPCHECK(hipEventRecord(ev[1], s));
    for (int l = 0; l < P.nLevels; l++) {
        const LevelPlan &L = c.levels[l];
        if (L.n_small) {
            tic(K_FRONT_SMALL);
            hipLaunchKernelGGL(k_front_small, dim3(L.n_small), dim3(TPB), L.small_lds, s, c.dp, c.d_tab.p + L.small_off, c.d_pool.p,
                               c.d_H.p, c.d_bad.p);
            toc();
        }
        if (L.n_med) {
            tic(K_FRONT_MEDIUM);
            hipLaunchKernelGGL(k_front_medium, dim3(L.n_med), dim3(TPB_MED), L.med_lds, s, c.dp, c.d_tab.p + L.med_off, c.d_pool.p,
                               c.d_H.p, c.d_bad.p);
            toc();
        }
        if (L.n_big) {
            tic(K_ASSEMBLE_BIG);
            hipLaunchKernelGGL(k_assemble_big, dim3(L.asm_big.grid), dim3(TPB), L.asm_lds, s, c.dp, c.d_tab.p + L.asm_big.list_off,
                               c.d_tab.p + L.asm_big.pre_off, L.asm_big.n, c.d_pool.p, c.d_H.p);
            toc();
            for (size_t k = 0; k < L.panel.size(); k++) {
                const Launch &pa = L.panel[k], &sy = L.syrk[k];
                tic(K_PANEL_BIG);
                hipLaunchKernelGGL(k_panel_big, dim3(pa.grid), dim3(TPB), 0, s, c.dp, c.d_tab.p + pa.list_off, c.d_tab.p + pa.pre_off,
                                   pa.n, (int)k, c.d_pool.p, c.d_bad.p);
                toc();
                if (sy.grid > 0) {
                    tic(K_SYRK_BIG);
                    hipLaunchKernelGGL(k_syrk_big, dim3(sy.grid), dim3(TPB), 0, s, c.dp, c.d_tab.p + sy.list_off, c.d_tab.p + sy.pre_off,
                                       sy.n, (int)k, c.d_pool.p);
                    toc();
                }
            }
        }
    }
    if (ev) HIPCHECK(hipEventRecord(ev[2], s));
    for (int l = P.nLevels - 1; l >= 0; l--) {
        const LevelPlan &L = c.levels[l];
        tic(K_BACKSOLVE);
        hipLaunchKernelGGL(k_backsolve, dim3(L.n_all), dim3(TPB), L.solve_lds, s, c.dp, c.d_tab.p + L.all_off, c.d_pool.p, c.d_x.p);
        toc();
    }
    HIPCHECK(hipMemsetAsync(gp.d_dx.p, 0xFF, (size_t)24 * N, s));     // NaN sentinel = "node skipped"
    tic(K_UPDATE);
    hipLaunchKernelGGL(k_update_states, dim3((N + TPB - 1) / TPB), dim3(TPB), 0, s, N, c.d_pos.p, c.d_x.p, gp.d_lp.p, gp.d_state.p, gp.d_dx.p);
    toc();
    if (ev) HIPCHECK(hipEventRecord(ev[3], s));
    HIPCHECK(hipGetLastError());
}
// after the stream was synchronised: fold the event pairs of the last instrumented enqueue into c.k_ms
static void collect_kernel_times(Context &c) {
    for (size_t i = 0; i < c.k_ids.size(); i++) {
        float ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, c.k_ev[2 * i], c.k_ev[2 * i + 1]));
        c.k_ms[c.k_ids[i]] += ms; c.k_calls[c.k_ids[i]]++;
    }
    c.k_ids.clear();
}

// run the numeric phase, replaying a captured hipGraph when enabled
static void run_numeric(Context &c, GraphPack &gp, bool timing, bool unary_at_lp = false) {
    hipStream_t s = gp.stream;
    set_small_attr();
    if (timing && !c.have_events) { for (auto &e : c.ev) HIPCHECK(hipEventCreate(&e)); c.have_events = true; }
    if (g_opt.use_graph && !timing && !unary_at_lp) {
        if (!c.gexec || c.gexec_key != (const void *)gp.d_state.p) {
            if (c.gexec) { (void)hipGraphExecDestroy(c.gexec); c.gexec = nullptr; }
            hipGraph_t graph = nullptr;
            HIPCHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            enqueue_numeric(c, gp, s, nullptr);
            HIPCHECK(hipStreamEndCapture(s, &graph));
            HIPCHECK(hipGraphInstantiate(&c.gexec, graph, nullptr, nullptr, 0));
            HIPCHECK(hipGraphDestroy(graph));
            c.gexec_key = (const void *)gp.d_state.p;
        }
        HIPCHECK(hipGraphLaunch(c.gexec, s));
    } else {
        enqueue_numeric(c, gp, s, timing ? c.ev : nullptr, unary_at_lp);
    }
}

static double device_chi2(GraphPack &gp) {     // chi^2 at d_state; synchronises the stream
    hipStream_t s = gp.stream;
    if (gp.F == 0) return 0;
    hipLaunchKernelGGL(k_chi2, dim3((gp.F + TPB - 1) / TPB), dim3(TPB), 0, s, gp.F, gp.d_fa.p, gp.d_fb.p, gp.d_z.p, gp.d_W.p, gp.d_state.p, gp.d_chi2f.p);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, s, gp.F, gp.d_chi2f.p, gp.d_scalar.p);
    HIPCHECK(hipMemcpyAsync(gp.h_scalar.p, gp.d_scalar.p, 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    return gp.h_scalar.p[0];
}

// make sure plan / device buffers match the packed graph; returns true if the plan was reused
static bool prepare_plan(Context &c, GraphPack &gp, const april_graph_t *g) {
    const int N = gp.N, F = gp.F;
    bool same = c.have_plan && c.patN == N && (int)c.pat.size() == 2 * F && c.plan.leaf_nodes == g_opt.leaf_nodes;
    if (same) {
        for (int i = 0; i < F && same; i++) same = c.pat[2 * i] == gp.h_fa.p[i] && c.pat[2 * i + 1] == gp.h_fb.p[i];
    }
    if (same) return true;
    c.pat.resize((size_t)2 * F);
    for (int i = 0; i < F; i++) { c.pat[2 * i] = gp.h_fa.p[i]; c.pat[2 * i + 1] = gp.h_fb.p[i]; }
    c.patN = N;
    std::vector<double> xy((size_t)2 * N);
    for (int i = 0; i < N; i++) { xy[2 * i] = gp.h_state.p[3 * i]; xy[2 * i + 1] = gp.h_state.p[3 * i + 1]; }
    build_plan(c.plan, N, F, c.pat.data(), xy.data(), g_opt.leaf_nodes);
    upload_plan(c, gp.stream);
    c.have_plan = true;
    return false;
}

static void set_lambda(Context &c, GraphPack &gp, double lambda) {
    const int N = c.plan.N;
    c.h_lambda.assign(N, lambda > 0 ? lambda : 0.0);            // aprilsam.c:197-204
    HIPCHECK(hipMemcpyAsync(c.d_lambda.p, c.h_lambda.data(), (size_t)8 * N, hipMemcpyHostToDevice, gp.stream));
}

// ------------------------------------------------------------------------------------------------------
// one batch Gauss-Newton step through the reference API (aprilsam.c:87-375)
// ------------------------------------------------------------------------------------------------------
static void batch_impl(april_graph_t *g, april_graph_cholesky_param_t *param) {
    Context &c = ctx_for(param);
    GraphPack &gp = pack_for(g);
    const double t0 = now_ms();
    pack_factors(gp, g);
    pack_states(gp, g, false);
    const int N = gp.N, F = gp.F;
    const double t1 = now_ms();
    const bool reused = prepare_plan(c, gp, g);
    const double t2 = now_ms();
    upload_factors(gp);
    // batch: every node is re-linearised first (aprilsam.c:131-135): l_point <- state, on the device
    HIPCHECK(hipMemcpyAsync(gp.d_lp.p, gp.d_state.p, (size_t)24 * N, hipMemcpyDeviceToDevice, gp.stream));
    set_lambda(c, gp, param->tikhanov);
    const double t3 = now_ms();
    const bool timing = g_opt.device_timing != 0;
    run_numeric(c, gp, timing);
    HIPCHECK(hipMemcpyAsync(gp.h_lp.p, gp.d_state.p, (size_t)24 * N, hipMemcpyDeviceToHost, gp.stream));   // new states
    HIPCHECK(hipMemcpyAsync(gp.h_dx.p, gp.d_dx.p, (size_t)24 * N, hipMemcpyDeviceToHost, gp.stream));
    HIPCHECK(hipMemcpyAsync(c.h_bad.p, c.d_bad.p, 4, hipMemcpyDeviceToHost, gp.stream));
    HIPCHECK(hipStreamSynchronize(gp.stream));
    const double t4 = now_ms();
    c.st.not_spd = c.h_bad.p[0] != 0;
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    if (c.st.not_spd) {
        static bool warned = false;
        if (!warned) { fprintf(stderr, "aprilsam_amd: information matrix not positive definite; node states left untouched\n"); warned = true; }
    } else {
        // write back: l_point = linearisation point used (old state), state/delta_X where not NaN-skipped,
        // UID = index (aprilsam.c:628)
        for (int i = N - 1; i >= 0; i--) {                                   // aprilsam.c:311-315 order
            april_graph_node_t *n = ns[i];
            n->UID = i;
            memcpy(n->l_point, gp.h_state.p + (size_t)3 * i, 24);
            const double *dx = gp.h_dx.p + (size_t)3 * i;
            if (std::isnan(dx[0]) || std::isnan(dx[1]) || std::isnan(dx[2])) continue;   // april_graph_xyt.c:304-305
            memcpy(n->state, gp.h_lp.p + (size_t)3 * i, 24);
            memcpy(n->delta_X, dx, 24);
        }
        // param bookkeeping the reference maintains (aprilsam.c:283-288)
        if (param->ordering) free(param->ordering);
        param->ordering = (int *)malloc(sizeof(int) * (size_t)N);
        memcpy(param->ordering, c.plan.perm.data(), sizeof(int) * (size_t)N);
        param->nreordering = N;
        param->factor_num = F;
        c.have_fact = true; c.batch_nodes = N; c.batch_factors = F; c.model.valid = false;
        if (param->delta_x) {                                                // aprilsam.c:363-366
            free(param->delta_x);
            param->delta_x = (double *)calloc((size_t)3 * N, sizeof(double));
            for (int i = 0; i < N; i++) memcpy(param->delta_x + (size_t)3 * c.plan.pos[i], gp.h_dx.p + (size_t)3 * i, 24);
        }
    }
    const double t5 = now_ms();
    c.st.n_nodes = N; c.st.n_factors = F; c.st.symbolic_reused = reused;
    c.st.ms_pack = t1 - t0; c.st.ms_symbolic = t2 - t1; c.st.ms_h2d = t3 - t2; c.st.ms_device = t4 - t3; c.st.ms_d2h = 0;
    c.st.ms_unpack = t5 - t4; c.st.ms_total = t5 - t0;
    if (timing) {
        float a = 0, b = 0, d = 0;
        HIPCHECK(hipEventElapsedTime(&a, c.ev[0], c.ev[1])); HIPCHECK(hipEventElapsedTime(&b, c.ev[1], c.ev[2]));
        HIPCHECK(hipEventElapsedTime(&d, c.ev[2], c.ev[3]));
        c.st.ms_dev_linearize = a; c.st.ms_dev_factor = b; c.st.ms_dev_solve = d;
    }
    if (param->show_timing) {
        printf("aprilsam_amd batch: N=%d F=%d fronts=%d levels=%d | pack %.3f symbolic %.3f%s upload %.3f device %.3f unpack %.3f | total %.3f ms\n",
               N, F, c.st.n_fronts, c.st.n_levels, c.st.ms_pack, c.st.ms_symbolic, reused ? " (cached)" : "", c.st.ms_h2d, c.st.ms_device,
               c.st.ms_unpack, c.st.ms_total);
    }
}

void batch_step(april_graph_t *g, april_graph_cholesky_param_t *param) {
    if (zsize(g->nodes) == 0 || zsize(g->factors) == 0) return;          // aprilsam.c:90-91
    if (!param->nreordering) fatal("april_graph_cholesky: param->nreordering == 0 (the reference asserts, aprilsam.c:372-374)");
    ensure_device();
    std::lock_guard<std::mutex> lk(g_mu);
    batch_impl(g, param);
}

// ------------------------------------------------------------------------------------------------------
// incremental step (aprilsam.c:377-576).  The linear system the reference maintains by partial un-/re-
// factorisation — every factor linearised at its nodes' l_point (aprilsam.c:508-542; l_points only move in
// a batch step), Tikhonov term only on poses present at the last batch step (aprilsam.c:197-204 vs :508-542)
// — is solved on the GPU (round 1: re-assembled and re-factorised in full); WHICH poses receive the result,
// the relinearisation counter and the batch fall-back follow the reference exactly through the bookkeeping
// model of refmodel.cpp (measured: on the poses it touches, the reference's result is the exact solution).
// ------------------------------------------------------------------------------------------------------
void inc_step(april_graph_t *g, april_graph_cholesky_param_t *param) {
    if (zsize(g->nodes) == 0 || zsize(g->factors) == 0) return;          // aprilsam.c:380-381
    std::lock_guard<std::mutex> lk(g_mu);
    {
        auto it = g_ctx.find(param);
        if (it == g_ctx.end() || !it->second->have_fact) return;         // aprilsam.c:382-383 (no prior chol)
    }
    if (param->factor_num == zsize(g->factors)) return;                  // aprilsam.c:384-385
    ensure_device();
    Context &c = ctx_for(param);
    GraphPack &gp = pack_for(g);
    const double t0 = now_ms();
    pack_factors(gp, g);
    pack_states(gp, g, true);
    const int N = gp.N, F = gp.F;
    if (!c.model.valid) c.model.batch(c.batch_nodes, c.batch_factors, gp.h_fa.p, gp.h_fb.p);      // lazily, after a batch step
    c.model.inc_begin(N, F, gp.h_fa.p, gp.h_fb.p);
    const bool reused = prepare_plan(c, gp, g);
    upload_factors(gp);
    c.h_lambda.assign(N, 0.0);
    for (int i = 0; i < N; i++) if (c.plan.perm[i] < c.batch_nodes && param->tikhanov > 0) c.h_lambda[i] = param->tikhanov;
    HIPCHECK(hipMemcpyAsync(c.d_lambda.p, c.h_lambda.data(), (size_t)8 * N, hipMemcpyHostToDevice, gp.stream));
    run_numeric(c, gp, false, true);
    HIPCHECK(hipMemcpyAsync(gp.h_state.p, gp.d_state.p, (size_t)24 * N, hipMemcpyDeviceToHost, gp.stream));
    HIPCHECK(hipMemcpyAsync(gp.h_dx.p, gp.d_dx.p, (size_t)24 * N, hipMemcpyDeviceToHost, gp.stream));
    HIPCHECK(hipMemcpyAsync(c.h_bad.p, c.d_bad.p, 4, hipMemcpyDeviceToHost, gp.stream));
    HIPCHECK(hipStreamSynchronize(gp.stream));
    c.st.not_spd = c.h_bad.p[0] != 0;
    c.st.n_nodes = N; c.st.n_factors = F; c.st.symbolic_reused = reused;
    if (c.st.not_spd) {
        static bool warned = false;
        if (!warned) { fprintf(stderr, "aprilsam_amd: incremental system not positive definite; node states left untouched\n"); warned = true; }
        return;
    }
    // bookkeeping exactly as the reference: which poses solve_node visits / updates, start_over (refmodel.cpp)
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    for (int i = 0; i < N; i++) ns[i]->UID = i;                          // aprilsam.c:474
    const double *x = gp.h_dx.p;                                          // dx per node; NaN where the solve produced NaN
    c.model.solve_visit(x, param->delta_xy, param->delta_theta, [&](int n, bool update) {
        april_graph_node_t *nd = ns[n];
        const double *dx = x + (size_t)3 * n;
        memcpy(nd->delta_X, dx, 24);                                      // aprilsam.c:752-754
        if (!update) return;
        if (std::isnan(dx[0]) || std::isnan(dx[1]) || std::isnan(dx[2])) return;   // april_graph_xyt.c:304-305
        memcpy(nd->state, gp.h_state.p + (size_t)3 * n, 24);              // l_point + dx, theta wrapped (k_update_states)
    });
    if (param->ordering) free(param->ordering);
    param->ordering = (int *)malloc(sizeof(int) * (size_t)N);
    memcpy(param->ordering, c.plan.perm.data(), sizeof(int) * (size_t)N);
    param->nreordering = N;
    param->factor_num = F;
    const double step_ms = now_ms() - t0;
    c.st.ms_total = step_ms;
    if (!g_opt.deterministic && step_ms > param->batch_time / 3) c.model.start_over = 0x7fffffff;    // aprilsam.c:557-559
    if (c.model.start_over > param->nthreshold) {                                                   // aprilsam.c:566-575
        const double b0 = now_ms();
        batch_impl(g, param);
        param->batch_time = now_ms() - b0;
    }
}

// aprilsam.c:578-597: back-substitution + state update on the current factorisation
void inc_solve_only(april_graph_t *g, april_graph_cholesky_param_t *param) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(param);
    if (it == g_ctx.end() || !it->second->have_fact || !param->nreordering) return;
    ensure_device();
    Context &c = *it->second;
    GraphPack &gp = pack_for(g);
    const Plan &P = c.plan;
    const int N = P.N;
    if (gp.N != N) return;
    hipStream_t s = gp.stream;
    set_small_attr();
    for (int l = P.nLevels - 1; l >= 0; l--) {
        const LevelPlan &L = c.levels[l];
        hipLaunchKernelGGL(k_backsolve, dim3(L.n_all), dim3(TPB), L.solve_lds, s, c.dp, c.d_tab.p + L.all_off, c.d_pool.p, c.d_x.p);
    }
    HIPCHECK(hipMemsetAsync(gp.d_dx.p, 0xFF, (size_t)24 * N, s));
    hipLaunchKernelGGL(k_update_states, dim3((N + TPB - 1) / TPB), dim3(TPB), 0, s, N, c.d_pos.p, c.d_x.p, gp.d_lp.p, gp.d_state.p, gp.d_dx.p);
    HIPCHECK(hipMemcpyAsync(gp.h_state.p, gp.d_state.p, (size_t)24 * N, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(gp.h_dx.p, gp.d_dx.p, (size_t)24 * N, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    for (int i = 0; i < N && i < zsize(g->nodes); i++) {
        const double *dx = gp.h_dx.p + (size_t)3 * i;
        if (std::isnan(dx[0]) || std::isnan(dx[1]) || std::isnan(dx[2])) continue;
        memcpy(ns[i]->state, gp.h_state.p + (size_t)3 * i, 24);
        memcpy(ns[i]->delta_X, dx, 24);
    }
}

double graph_chi2(april_graph_t *g) {
    if (zsize(g->factors) == 0) return 0;
    ensure_device();
    std::lock_guard<std::mutex> lk(g_mu);
    GraphPack &gp = pack_for(g);
    pack_factors(gp, g);
    pack_states(gp, g, false);
    upload_factors(gp);
    return device_chi2(gp);
}

// ------------------------------------------------------------------------------------------------------
// device-resident driver API: states never leave HBM between Gauss-Newton steps
// ------------------------------------------------------------------------------------------------------
int resident_begin(april_graph_t *g, april_graph_cholesky_param_t *param) {
    if (zsize(g->nodes) == 0 || zsize(g->factors) == 0) return -1;
    ensure_device();
    std::lock_guard<std::mutex> lk(g_mu);
    Context &c = ctx_for(param);
    GraphPack &gp = pack_for(g);
    pack_factors(gp, g);
    pack_states(gp, g, false);
    const bool reused = prepare_plan(c, gp, g);
    upload_factors(gp);
    set_lambda(c, gp, param->tikhanov);
    if (!c.have_events) { for (auto &e : c.ev) HIPCHECK(hipEventCreate(&e)); c.have_events = true; }
    for (int k = 0; k < NKERN; k++) { c.k_ms[k] = 0; c.k_calls[k] = 0; }
    c.st.n_nodes = gp.N; c.st.n_factors = gp.F; c.st.symbolic_reused = reused; c.st.not_spd = 0;
    HIPCHECK(hipStreamSynchronize(gp.stream));
    return 0;
}
// enqueue n iterations.  mode 0: asynchronous (hipGraph replay when enabled), returns at once;
// mode 1: every kernel bracketed by HIP events on the solver stream, synchronises after each iteration.
int resident_steps(april_graph_t *g, april_graph_cholesky_param_t *param, int n, int mode) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(param);
    if (it == g_ctx.end() || !it->second->have_plan) return -1;
    Context &c = *it->second;
    GraphPack &gp = pack_for(g);
    hipStream_t s = gp.stream;
    const int N = gp.N;
    HIPCHECK(hipSetDevice(g_device));
    set_small_attr();
    for (int i = 0; i < n; i++) {
        HIPCHECK(hipMemcpyAsync(gp.d_lp.p, gp.d_state.p, (size_t)24 * N, hipMemcpyDeviceToDevice, s));   // relinearise
        if (mode == 1) {
            enqueue_numeric(c, gp, s, nullptr, false, true);
            HIPCHECK(hipStreamSynchronize(s));
            collect_kernel_times(c);
        } else {
            run_numeric(c, gp, false);
        }
    }
    return 0;
}
int resident_sync(april_graph_t *g, april_graph_cholesky_param_t *param) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(param);
    if (it == g_ctx.end()) return -1;
    Context &c = *it->second;
    GraphPack &gp = pack_for(g);
    HIPCHECK(hipMemcpyAsync(c.h_bad.p, c.d_bad.p, 4, hipMemcpyDeviceToHost, gp.stream));
    HIPCHECK(hipStreamSynchronize(gp.stream));
    c.st.not_spd = c.h_bad.p[0] != 0;
    return c.h_bad.p[0] ? -2 : 0;
}
double resident_chi2(april_graph_t *g) {
    std::lock_guard<std::mutex> lk(g_mu);
    return device_chi2(pack_for(g));
}
int resident_end(april_graph_t *g, april_graph_cholesky_param_t *param) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(param);
    if (it == g_ctx.end()) return -1;
    Context &c = *it->second;
    GraphPack &gp = pack_for(g);
    hipStream_t s = gp.stream;
    const int N = gp.N, F = gp.F;
    HIPCHECK(hipMemcpyAsync(gp.h_state.p, gp.d_state.p, (size_t)24 * N, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(gp.h_lp.p, gp.d_lp.p, (size_t)24 * N, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(gp.h_dx.p, gp.d_dx.p, (size_t)24 * N, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    for (int i = 0; i < N; i++) {
        april_graph_node_t *n = ns[i];
        n->UID = i;
        memcpy(n->state, gp.h_state.p + (size_t)3 * i, 24);
        memcpy(n->l_point, gp.h_lp.p + (size_t)3 * i, 24);
        const double *dx = gp.h_dx.p + (size_t)3 * i;
        if (!(std::isnan(dx[0]) || std::isnan(dx[1]) || std::isnan(dx[2]))) memcpy(n->delta_X, dx, 24);
    }
    if (param->ordering) free(param->ordering);
    param->ordering = (int *)malloc(sizeof(int) * (size_t)N);
    memcpy(param->ordering, c.plan.perm.data(), sizeof(int) * (size_t)N);
    param->nreordering = N; param->factor_num = F;
    c.have_fact = true; c.batch_nodes = N; c.batch_factors = F; c.model.valid = false;
    return 0;
}
int batch_resident(april_graph_t *g, april_graph_cholesky_param_t *param, int iters, double *chi2_out, double *ms_out) {
    int rc = resident_begin(g, param);
    if (rc) return rc;
    if (chi2_out) chi2_out[0] = resident_chi2(g);
    for (int it = 0; it < iters && rc == 0; it++) {
        const double t0 = now_ms();
        resident_steps(g, param, 1, 0);
        rc = resident_sync(g, param);
        if (ms_out) ms_out[it] = now_ms() - t0;
        if (rc == 0 && chi2_out) chi2_out[it + 1] = resident_chi2(g);
    }
    if (rc == 0) rc = resident_end(g, param);
    return rc;
}
// per-kernel profile of the instrumented passes since resident_begin + algorithmic work per ITERATION
int kernel_profile(const april_graph_cholesky_param_t *param, double *ms, long long *calls, double *flops, double *bytes, const char **names) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(param);
    if (it == g_ctx.end() || !it->second->have_plan) return -1;
    Context &c = *it->second;
    const Plan &P = c.plan;
    for (int k = 0; k < NKERN; k++) { ms[k] = c.k_ms[k]; calls[k] = c.k_calls[k]; flops[k] = 0; bytes[k] = 0; if (names) names[k] = KNAMES[k]; }
    const size_t small_max = (size_t)g_opt.small_lds_kb * 1024, med_max = (size_t)g_opt.medium_lds_kb * 1024;
    for (int t = 0; t < P.nF; t++) {
        const double ns = 3.0 * P.f_nsb[t], nu = 3.0 * P.f_nub[t], R = P.rows(t), C = P.cols(t);
        double fl = 0;                                          // sum_j c_j^2 over this front's columns (+ rhs row)
        for (int q = 0; q < (int)ns; q++) { double cj = (ns - q) + nu + 1; fl += cj * cj; }
        const bool small = small_front_lds((int)R, (int)C) <= small_max, medium = !small && medium_front_lds((int)R) <= med_max;
        // algorithmic bytes of a front: its L panel + update block written once, children's updates read once
        const double by = 8.0 * (ns * (ns + 1) / 2 + (nu + 1) * ns + (nu + 1) * (nu + 2) / 2);
        if (small) { flops[K_FRONT_SMALL] += fl; bytes[K_FRONT_SMALL] += by; }
        else if (medium) { flops[K_FRONT_MEDIUM] += fl; bytes[K_FRONT_MEDIUM] += by; }
        else { flops[K_SYRK_BIG] += fl; bytes[K_SYRK_BIG] += by; }
        bytes[K_BACKSOLVE] += 8.0 * (ns * (ns + 1) / 2 + nu * ns) + 16.0 * (ns + nu);
        flops[K_BACKSOLVE] += 2.0 * (ns * (ns + 1) / 2 + nu * ns);
    }
    // SURVEY.md section 8(d) assembly bytes: factor records + poses read, contribution blocks written
    int F2 = 0, F1 = 0;
    for (int f = 0; f < P.F; f++) (c.pat[2 * f + 1] >= 0 ? F2 : F1)++;
    bytes[K_LINEARIZE] = F2 * 152.0 + F1 * 124.0 + 8.0 * (27.0 * F2 + 9.0 * F1 + 6.0 * F2 + 3.0 * F1);
    flops[K_LINEARIZE] = 150.0 * F2 + 40.0 * F1;
    bytes[K_UPDATE] = 8.0 * 3 * P.N * 4;
    return NKERN;
}

// debug: copy the per-front clock stamps (8 per front) written when APRILSAM_AMD_KPROF is set
int debug_front_times(const april_graph_cholesky_param_t *param, long long *out, int n_fronts) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(param);
    if (it == g_ctx.end() || !it->second->d_prof.p) return -1;
    HIPCHECK(hipDeviceSynchronize());
    int n = std::min(n_fronts, it->second->plan.nF);
    HIPCHECK(hipMemcpy(out, it->second->d_prof.p, (size_t)64 * n, hipMemcpyDeviceToHost));
    return n;
}
int api_device_count() { return device_count(); }
int api_set_device(int d) {
    int n = device_count();
    if (d < 0 || d >= n) return -1;
    g_device = d;
    return 0;
}
int api_set_option(const char *name, double v) {
    load_env_options();
    std::string k(name);
    if (k == "leaf_nodes") g_opt.leaf_nodes = (int)v;
    else if (k == "deterministic") g_opt.deterministic = (int)v;
    else if (k == "use_graph") g_opt.use_graph = (int)v;
    else if (k == "device_timing") g_opt.device_timing = (int)v;
    else if (k == "trust_factor_cache") g_opt.trust_factor_cache = (int)v;
    else if (k == "small_lds_kb") g_opt.small_lds_kb = (int)v;
    else if (k == "medium_lds_kb") g_opt.medium_lds_kb = (int)v;
    else return -1;
    return 0;
}

}  // namespace asam

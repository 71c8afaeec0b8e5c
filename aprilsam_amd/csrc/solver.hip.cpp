// solver.hip.cpp — host runtime of the MI355X Gauss-Newton step: packs the caller's graph objects
// (reference ABI, include/aprilsam_amd.h PART 1) into SoA arrays, keeps ordering / symbolic plan /
// HBM-resident fronts in a side context keyed by the param pointer, and drives the HIP kernels on one
// stream per context.  Entry points (C ABI) are at the bottom.
//
// Reference call stack this replaces: aprilsam.c:87-375 (april_graph_cholesky) — see SURVEY.md §3.1.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/aprilsam_amd.h"
#include "errors.h"
#include "kernels.hip.h"
#include "plan.h"
#include "refmodel.h"
#include "solver.h"

namespace asam {

// ------------------------------------------------------------------------------------------------------
// utilities
// ------------------------------------------------------------------------------------------------------
// every failure is a SolverError caught at the entry point (guarded() below): states untouched, message on stderr,
// code kept for aprilsam_amd_last_error / stats.error_code -- "no HIP device" included (ERR_NO_DEVICE): there is no CPU
// fallback to fall back to, every solver call on such a box fails, loudly, and returns (round 4: no abort() left in the library)
#define HIPCHECK(expr)                                                                                 \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (void)hipGetLastError();                                                                   \
            fail(e_ == hipErrorOutOfMemory ? ERR_OOM : ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
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
        v = g_opt.syrk128_rows; envd("APRILSAM_AMD_SYRK128_ROWS", &v); g_opt.syrk128_rows = (int)v;
        v = g_opt.panel_mode; envd("APRILSAM_AMD_PANEL_MODE", &v); g_opt.panel_mode = (int)v;
        v = g_opt.small_threads; envd("APRILSAM_AMD_SMALL_THREADS", &v); g_opt.small_threads = (int)v;
        v = g_opt.tp_fronts; envd("APRILSAM_AMD_TP_FRONTS", &v); g_opt.tp_fronts = (int)v;
        v = g_opt.tp_lds_kb; envd("APRILSAM_AMD_TP_LDS_KB", &v); g_opt.tp_lds_kb = (int)v;
        v = g_opt.tp_threads; envd("APRILSAM_AMD_TP_THREADS", &v); g_opt.tp_threads = (int)v;
        v = g_opt.lookahead; envd("APRILSAM_AMD_LOOKAHEAD", &v); g_opt.lookahead = (int)v;
        v = g_opt.inc_fast; envd("APRILSAM_AMD_INC_FAST", &v); g_opt.inc_fast = (int)v;
        v = g_opt.inc_multi; envd("APRILSAM_AMD_INC_MULTI", &v); g_opt.inc_multi = (int)v;
        v = g_opt.inc_one; envd("APRILSAM_AMD_INC_ONE", &v); g_opt.inc_one = (int)v;
        v = g_opt.inc_one_up; envd("APRILSAM_AMD_INC_ONE_UP", &v); g_opt.inc_one_up = (int)v;
        v = g_opt.inc_one_dn; envd("APRILSAM_AMD_INC_ONE_DN", &v); g_opt.inc_one_dn = (int)v;
        v = g_opt.inc_one_threads; envd("APRILSAM_AMD_INC_ONE_THREADS", &v); g_opt.inc_one_threads = (int)v;
        v = g_opt.inc_one_spin; envd("APRILSAM_AMD_INC_ONE_SPIN", &v); g_opt.inc_one_spin = (int)v;
        v = g_opt.inc_tail; envd("APRILSAM_AMD_INC_TAIL", &v); g_opt.inc_tail = (int)v;
        v = g_opt.inc_inline; envd("APRILSAM_AMD_INC_INLINE", &v); g_opt.inc_inline = (int)v;
        v = g_opt.inc_update; envd("APRILSAM_AMD_INC_UPDATE", &v); g_opt.inc_update = (int)v;
        v = g_opt.inc_tail_solve; envd("APRILSAM_AMD_INC_TAIL_SOLVE", &v); g_opt.inc_tail_solve = (int)v;
        v = g_opt.inc_lazy_states; envd("APRILSAM_AMD_INC_LAZY_STATES", &v); g_opt.inc_lazy_states = (int)v;
        v = g_opt.inc_replan_tall; envd("APRILSAM_AMD_INC_REPLAN_TALL", &v); g_opt.inc_replan_tall = (int)v;
        v = g_opt.speculate_factors; envd("APRILSAM_AMD_SPECULATE_FACTORS", &v); g_opt.speculate_factors = (int)v;
        v = g_opt.block_factor; envd("APRILSAM_AMD_BLOCK_FACTOR", &v); g_opt.block_factor = (int)v;
        v = g_opt.fused_panel; envd("APRILSAM_AMD_FUSED_PANEL", &v); g_opt.fused_panel = (int)v;
        v = g_opt.persist; envd("APRILSAM_AMD_PERSIST", &v); g_opt.persist = (int)v;
        v = g_opt.persist_max_fronts; envd("APRILSAM_AMD_PERSIST_MAX_FRONTS", &v); g_opt.persist_max_fronts = (int)v;
        v = g_opt.linearize_staged_min; envd("APRILSAM_AMD_LINEARIZE_STAGED_MIN", &v); g_opt.linearize_staged_min = (int)v;
        v = g_opt.wave_backsolve; envd("APRILSAM_AMD_WAVE_BACKSOLVE", &v); g_opt.wave_backsolve = (int)v;
        v = g_opt.left_panels; envd("APRILSAM_AMD_LEFT_PANELS", &v); g_opt.left_panels = (int)v;
        v = g_opt.block_panels; envd("APRILSAM_AMD_BLOCK_PANELS", &v); g_opt.block_panels = (int)v;
        v = g_opt.blk_backsolve; envd("APRILSAM_AMD_BLK_BACKSOLVE", &v); g_opt.blk_backsolve = (int)v;
        v = g_opt.tile_assembly; envd("APRILSAM_AMD_TILE_ASSEMBLY", &v); g_opt.tile_assembly = (int)v;
        v = g_opt.tail_poses; envd("APRILSAM_AMD_TAIL_POSES", &v); g_opt.tail_poses = std::max(8, (int)v);
        v = g_opt.batch_extend; envd("APRILSAM_AMD_BATCH_EXTEND", &v); g_opt.batch_extend = (int)v;
        v = g_opt.extend_tail_fronts; envd("APRILSAM_AMD_EXTEND_TAIL_FRONTS", &v); g_opt.extend_tail_fronts = (int)v;
        v = g_opt.mem_cap_mb; envd("APRILSAM_AMD_MEM_CAP_MB", &v); g_opt.mem_cap_mb = (int)v;
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
    std::call_once(once, [] {           // (an exception leaves the flag unset: the next call looks again)
        load_env_options();
        int n = device_count();
        if (n <= 0) fail(ERR_NO_DEVICE, "no HIP device visible: the april_graph_cholesky* / april_graph_chi2 entry points of "
                         "libaprilsam_amd.so run on an AMD GPU only (there is NO CPU fallback: nothing was computed)");
        if (g_device < 0) {
            const char *lr = getenv("LOCAL_RANK");
            g_device = lr ? atoi(lr) % n : 0;
        }
    });
    HIPCHECK(hipSetDevice(g_device));
}

// grow-only device / pinned-host buffers.  A failed allocation leaves the buffer empty (never dangling) and throws ERR_OOM;
// option mem_cap_mb (0 = off) refuses any single device buffer above that size the same way -- the tests use it to walk the
// out-of-memory path without exhausting a 288 GB device.
template <class T> struct DBuf {
    T *p = nullptr; size_t cap = 0;
    void need(size_t n) {
        if (n <= cap) return;
        size_t c = std::max(n, cap + cap / 2);
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }       // (freed first: the front pool of a large graph must not exist twice)
        if (g_opt.mem_cap_mb > 0 && n * sizeof(T) > ((size_t)g_opt.mem_cap_mb << 20))
            fail(ERR_OOM, "device buffer of %.1f MB refused: option mem_cap_mb = %d", (double)(n * sizeof(T)) / 1048576.0, g_opt.mem_cap_mb);
        hipError_t e = hipMalloc((void **)&p, c * sizeof(T));
        if (e != hipSuccess && c > n) { (void)hipGetLastError(); c = n; e = hipMalloc((void **)&p, c * sizeof(T)); }      // without the head-room
        if (e != hipSuccess) {
            (void)hipGetLastError(); p = nullptr;
            fail(e == hipErrorOutOfMemory ? ERR_OOM : ERR_HIP, "hipMalloc of %.1f MB failed: %s", (double)(c * sizeof(T)) / 1048576.0, hipGetErrorString(e));
        }
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
        const hipError_t e = hipHostMalloc((void **)&q, c * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) {           // the old buffer stays valid
            (void)hipGetLastError();
            fail(e == hipErrorOutOfMemory ? ERR_OOM : ERR_HIP, "hipHostMalloc of %.1f MB failed: %s", (double)(c * sizeof(T)) / 1048576.0, hipGetErrorString(e));
        }
        if (p) { if (keep) memcpy(q, p, cap * sizeof(T)); (void)hipHostFree(p); }
        p = q; cap = c;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// staging of the small per-step table updates of the incremental path: see k_apply_patches
struct PatchList {
    HBuf<char> buf; std::vector<Patch> hdr; size_t used = 0;
    void reset() { hdr.clear(); used = 0; }
    void add(void *dst, const void *src, size_t bytes) {
        if (!bytes) return;
        const size_t off = (used + 15) & ~(size_t)15;
        buf.need(off + bytes + 64, true);
        memcpy(buf.p + off, src, bytes);
        hdr.push_back(Patch{ dst, (long long)off, (long long)bytes });
        used = off + bytes;
    }
    // appends the header behind the payload; returns it (pinned memory the kernels read across PCIe)
    const Patch *finish() {
        const size_t hoff = (used + 63) & ~(size_t)63, hbytes = hdr.size() * sizeof(Patch);
        buf.need(hoff + hbytes + 64, true);
        memcpy(buf.p + hoff, hdr.data(), hbytes);
        return (const Patch *)(buf.p + hoff);
    }
    // ... and launches the scatter kernel (nothing to do: no launch)
    void launch(hipStream_t s) {
        if (hdr.empty()) return;
        const Patch *h = finish();
        hipLaunchKernelGGL(k_apply_patches, dim3((unsigned)hdr.size()), dim3(TPB), 0, s, h, (const char *)buf.p);
    }
    void release() { buf.release(); hdr.clear(); used = 0; }
};

// ------------------------------------------------------------------------------------------------------
// packed graph (SoA, host pinned + device) — one per april_graph_t pointer
// ------------------------------------------------------------------------------------------------------
static long long g_pack_serial = 0;
struct GraphPack {
    const long long serial = ++g_pack_serial;      // captured hipGraphs are keyed by it: a pack freed and another allocated at the same addresses must not match
    int N = 0, F = 0;                  // packed counts (F: packed factor entries, see pack_factors)
    int Fg = 0;                        // graph factors packed (== F unless a factor has more than two nodes)
    std::vector<int> g2p, p2g;         // graph factor -> its first packed entry (size Fg + 1) / packed entry -> graph factor
    std::vector<unsigned> vslot;       // per packed entry of a host-evaluated factor: node slots (x << 8 | y; y = 0xff: unary) | carry bits << 16
    std::vector<const void *> fptr;    // factor object pointers already packed (cache validation), one per graph factor
    std::vector<int> pending;          // poses whose pinned state mirror is ahead of the device copy (written by apply_visits)
    HBuf<int> h_fa, h_fb;
    HBuf<double> h_z, h_W, h_state, h_lp, h_dx;
    DBuf<int> d_fa, d_fb;
    DBuf<double> d_z, d_W, d_state, d_lp, d_dx, d_chi2f, d_scalar;
    int F_on_device = 0;               // factors already uploaded
    int dirty_lo = 0, dirty_hi = 0;    // packed factors whose z / W changed since the last upload
    long long content_version = 0;     // bumped whenever z / W of a packed factor changed
    std::vector<char> is_host;         // per factor: evaluated on the host through factor->eval
    std::vector<double> h_upt; DBuf<double> d_upt;   // unary factors: the state they were linearised at when they entered the system (3 per factor)
    int F_cap = 0;                     // device capacity (factors) of d_fa/d_fb/d_z/d_W/d_chi2f
    // factors of foreign types, evaluated on the host through factor->eval: indices, 33 doubles each
    // (Haa, Hab, Hbb, ga, gb), how many of them hold a current evaluation
    std::vector<int> host_idx; HBuf<double> h_hostH; DBuf<double> d_hostH; DBuf<int> d_host_idx; int host_evaluated = 0;
    hipStream_t stream = nullptr;
    HBuf<double> h_scalar;
    // incremental steps: the pinned mirrors h_state / h_lp and the device arrays d_state / d_lp hold the same values (mirror_sync),
    // so a step only has to patch the poses whose host objects differ from the mirror (pack_states_diff); the step's new
    // states go to h_out (pinned), not into the mirror
    HBuf<double> h_out; bool mirror_sync = false; std::vector<int> changed; const double *new_states = nullptr;
    void release() {
        h_out.release(); mirror_sync = false;
        h_fa.release(); h_fb.release(); h_z.release(); h_W.release(); h_state.release(); h_lp.release(); h_dx.release();
        d_fa.release(); d_fb.release(); d_z.release(); d_W.release(); d_state.release(); d_lp.release(); d_dx.release();
        d_chi2f.release(); d_scalar.release(); h_scalar.release(); h_hostH.release(); d_hostH.release(); d_host_idx.release(); d_upt.release();
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

// (re)pack the factors: ids, z, W into the pinned SoA mirror.  The reference re-reads every factor object on every call
// (aprilsam.c:152-190, april_graph.c:79-98), so by default every already-packed factor is compared with the mirror
// (nodes, z, W: 104 bytes) and only what changed is copied and uploaded again; a changed endpoint or factor kind
// restarts the pack.  Option trust_factor_cache = 1 skips the comparison for factors whose object pointer is unchanged
// (z / W of a packed factor are then treated as immutable).
//
// PACKED factors are what everything below this function sees: one entry per graph factor with one or two nodes, and for a
// factor with k >= 3 nodes (foreign types only, evaluated through their own eval(): the reference's assembly loops are
// generic over factor->nnodes, aprilsam.c:159-192) one entry per PAIR of its nodes, k (k - 1) / 2 of them -- the pair (i, j)
// carries the off-diagonal block J_i^T W J_j; the diagonal block and the right-hand-side segment of node i ride on the
// first pair that contains i.  A clique of binary entries is exactly the structure such a factor has in the normal
// equations, so ordering, symbolic analysis and kernels need not know.  gp.F counts packed entries, gp.Fg graph factors
// (param->factor_num, aprilsam.c:283-288); g2p / p2g translate.
static void pack_factors(GraphPack &gp, const april_graph_t *g, bool validate_old = true) {
    const bool trust = g_opt.trust_factor_cache || !validate_old;      // (incremental calls never re-read old factors, aprilsam.c:508-511)
    const int Fg = zsize(g->factors);
    april_graph_factor_t **fs = (april_graph_factor_t **)g->factors->data;
    const int N = zsize(g->nodes);
    int from = gp.Fg;
    bool valid = from <= Fg && (int)gp.fptr.size() == from && (int)gp.g2p.size() == from + 1;
    // (incremental calls only ever look at the factors added since the previous call, aprilsam.c:508-511: first and last packed pointer
    // as a sanity check instead of all of them -- the comparison of 5 000 pointers was a microsecond of every step)
    if (valid && trust) valid = from == 0 || (validate_old ? memcmp(gp.fptr.data(), fs, sizeof(void *) * from) == 0 : (gp.fptr[0] == fs[0] && gp.fptr[from - 1] == fs[from - 1]));
    auto restart = [&]() { from = 0; gp.F = 0; gp.F_on_device = 0; gp.host_idx.clear(); gp.host_evaluated = 0; gp.is_host.clear(); gp.p2g.clear(); gp.vslot.clear(); gp.g2p.assign(1, 0); };
    if (!valid) restart();
    // one graph factor -> its packed entries (a, b, host flag, node slots of a host pair, what the pair carries)
    struct Ent { int a, b; bool host; unsigned short slots; unsigned char carry; };
    Ent ents[64]; int ne = 0;
    auto classify = [&](const april_graph_factor_t *f, int i) {
        ne = 0;
        if (f->type == APRIL_GRAPH_FACTOR_XYT_TYPE && f->nnodes == 2) ents[ne++] = Ent{ f->nodes[0], f->nodes[1], false, 0, 3 };
        else if (f->type == APRIL_GRAPH_FACTOR_XYTPOS_TYPE && f->nnodes == 1) ents[ne++] = Ent{ f->nodes[0], -1, false, 0, 3 };
        else if ((f->nnodes == 1 || f->nnodes == 2) && f->eval)       // any other type: the factor's own eval(), on the host
            ents[ne++] = Ent{ f->nodes[0], f->nnodes == 2 ? f->nodes[1] : -1, true, (unsigned short)(f->nnodes == 2 ? 1 : 0xff), 3 };
        else if (f->nnodes >= 3 && f->nnodes <= 11 && f->eval) {      // a clique of pairs (see above); 11 nodes = 55 pairs
            for (int x = 0; x < f->nnodes; x++)
                for (int y = x + 1; y < f->nnodes; y++) {
                    // node x's diagonal block / rhs on its first pair: (0, 1) for x = 0 and x = 1, (0, x) beyond
                    const unsigned char carry = (unsigned char)(((x == 0 && y == 1) ? 1 : 0) | ((x == 0) ? 2 : 0));
                    ents[ne++] = Ent{ f->nodes[x], f->nodes[y], true, (unsigned short)((x << 8) | y), carry };
                }
        } else {
            fail(ERR_UNSUPPORTED, "factor %d has type %d / %d nodes; factors of foreign types are supported with 1 to 11 nodes and an "
                                  "eval() function pointer (aprilsam.h:110-122)", i, f->type, f->nnodes);
        }
        for (int e = 0; e < ne; e++) {
            if (ents[e].a < 0 || ents[e].a >= N || ents[e].b >= N) fail(ERR_BAD_GRAPH, "factor %d references node %d / %d of %d", i, ents[e].a, ents[e].b, N);
            if (ents[e].a == ents[e].b) fail(ERR_BAD_GRAPH, "factor %d connects node %d to itself", i, ents[e].a);
        }
    };
    if (from > 0 && !trust) {
        // content check of the packed prefix; dirty range [lo, hi) is uploaded again by upload_factors
        int lo = gp.F, hi = 0;
        bool changed = false;
        for (int i = 0; i < from; i++) {
            if (i + 8 < from) __builtin_prefetch(fs[i + 8]);
            const april_graph_factor_t *f = fs[i];
            classify(f, i);
            const int p0 = gp.g2p[i];
            if (gp.g2p[i + 1] - p0 != ne) { changed = true; break; }
            for (int e = 0; e < ne && !changed; e++)
                changed = ents[e].a != gp.h_fa.p[p0 + e] || ents[e].b != gp.h_fb.p[p0 + e] || ents[e].host != (bool)gp.is_host[p0 + e];
            if (changed) break;
            gp.fptr[i] = f;
            if (ents[0].host) continue;
            double *zp = gp.h_z.p + (size_t)3 * p0, *Wp = gp.h_W.p + (size_t)9 * p0;
            if (memcmp(zp, f->u.common.z, 24) != 0 || memcmp(Wp, f->u.common.W->data, 72) != 0) {
                memcpy(zp, f->u.common.z, 24); memcpy(Wp, f->u.common.W->data, 72);
                lo = std::min(lo, p0); hi = std::max(hi, p0 + 1);
            }
        }
        if (changed) restart();
        else if (hi > lo) {
            if (gp.dirty_hi > gp.dirty_lo) { gp.dirty_lo = std::min(gp.dirty_lo, lo); gp.dirty_hi = std::max(gp.dirty_hi, hi); }
            else { gp.dirty_lo = lo; gp.dirty_hi = hi; }
            gp.content_version++;
        }
    }
    gp.fptr.resize(Fg); gp.g2p.resize((size_t)Fg + 1);
    int F = gp.g2p[from];
    for (int i = from; i < Fg; i++) {
        const april_graph_factor_t *f = fs[i];
        gp.fptr[i] = f;
        classify(f, i);
        gp.h_fa.need((size_t)F + ne, true); gp.h_fb.need((size_t)F + ne, true); gp.h_z.need((size_t)3 * (F + ne), true); gp.h_W.need((size_t)9 * (F + ne), true);
        gp.is_host.resize((size_t)F + ne, 0); gp.p2g.resize((size_t)F + ne); gp.vslot.resize((size_t)F + ne);
        for (int e = 0; e < ne; e++, F++) {
            gp.h_fa.p[F] = ents[e].a; gp.h_fb.p[F] = ents[e].b; gp.is_host[F] = ents[e].host; gp.p2g[F] = i;
            gp.vslot[F] = (unsigned)ents[e].slots | ((unsigned)ents[e].carry << 16);
            if (ents[e].host) {          // the device kernels see a null factor (W = 0) in its place; k_scatter_host fills its slots
                memset(gp.h_z.p + (size_t)3 * F, 0, 24); memset(gp.h_W.p + (size_t)9 * F, 0, 72);
                gp.host_idx.push_back(F);
            } else {
                memcpy(gp.h_z.p + (size_t)3 * F, f->u.common.z, 24);
                memcpy(gp.h_W.p + (size_t)9 * F, f->u.common.W->data, 72);
            }
        }
        gp.g2p[i + 1] = F;
    }
    gp.F = F; gp.Fg = Fg;
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
    const int d0 = gp.dirty_lo, d1 = std::min(gp.dirty_hi, f0);      // z / W of packed factors edited in place by the caller
    if (d1 > d0) {
        HIPCHECK(hipMemcpyAsync(gp.d_z.p + (size_t)3 * d0, gp.h_z.p + (size_t)3 * d0, (size_t)(d1 - d0) * 24, hipMemcpyHostToDevice, gp.stream));
        HIPCHECK(hipMemcpyAsync(gp.d_W.p + (size_t)9 * d0, gp.h_W.p + (size_t)9 * d0, (size_t)(d1 - d0) * 72, hipMemcpyHostToDevice, gp.stream));
    }
    gp.dirty_lo = gp.dirty_hi = 0;
    gp.F_on_device = F;
    gp.d_scalar.need(8); gp.h_scalar.need(8);
}
// evaluate the host factors [from, end) through their vtable (aprilsam.c:156 calls factor->eval the same way) and form
// (J_a^T W) J_a, (J_a^T W) J_b, (J_b^T W) J_b, (J^T W) r in the reference's association (aprilsam.c:162-187)
static double eval_host_factors(GraphPack &gp, april_graph_t *g, int from) {
    const int nh = (int)gp.host_idx.size();
    gp.h_hostH.need((size_t)33 * std::max(nh, 1), true);
    april_graph_factor_t **fs = (april_graph_factor_t **)g->factors->data;
    double chi2 = 0;
    april_graph_factor_eval_t *e = nullptr; int e_of = -1;         // (the pairs of a factor with more than two nodes share one evaluation)
    std::vector<double> JtW;
    for (int k = from; k < nh; k++) {
        const int hp = gp.host_idx[k], gi = gp.p2g[hp];
        april_graph_factor_t *f = fs[gi];
        const int x = (int)((gp.vslot[hp] >> 8) & 0xff), y = (int)(gp.vslot[hp] & 0xff), carry = (int)(gp.vslot[hp] >> 16);
        if (gi != e_of) {
            if (e) april_graph_factor_eval_destroy(e);
            e = f->eval(f, g, nullptr); e_of = gi;
            if (!e || !e->jacobians || !e->jacobians[0] || !e->W || !e->r) fail(ERR_BAD_GRAPH, "factor->eval returned an incomplete evaluation (aprilsam.h:75-89)");
            chi2 += e->chi2;
        }
        const int L = e->length;
        double *H = gp.h_hostH.p + (size_t)33 * k;
        memset(H, 0, 33 * 8);
        JtW.resize((size_t)3 * L);
        const int zs[2] = { x, y == 0xff ? -1 : y };
        for (int s0 = 0; s0 < 2; s0++) {
            const int z0 = zs[s0];
            if (z0 < 0) continue;
            const matd_t *J0 = e->jacobians[z0];
            if (!J0) fail(ERR_BAD_GRAPH, "factor->eval: fewer jacobians than nodes");
            if ((int)J0->nrows != L || J0->ncols != 3 || (int)e->W->nrows != L || (int)e->W->ncols != L)
                fail(ERR_UNSUPPORTED, "factor->eval: jacobians must be length x 3 and W length x length (3-DoF xyt nodes only, aprilsam.c:617)");
            for (int i = 0; i < 3; i++)
                for (int l = 0; l < L; l++) { double acc = 0; for (int m = 0; m < L; m++) acc += J0->data[m * 3 + i] * e->W->data[m * L + l]; JtW[(size_t)i * L + l] = acc; }
            for (int s1 = s0; s1 < 2; s1++) {
                const int z1 = zs[s1];
                if (z1 < 0) continue;
                if (s1 == s0 && !((carry >> s0) & 1)) continue;      // this node's diagonal block rides on another pair of the factor
                const matd_t *J1 = e->jacobians[z1];
                if (!J1) fail(ERR_BAD_GRAPH, "factor->eval: fewer jacobians than nodes");
                double *B = H + (s0 == 0 ? (s1 == 0 ? 0 : 9) : 18);
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) { double acc = 0; for (int l = 0; l < L; l++) acc += JtW[(size_t)i * L + l] * J1->data[l * 3 + j]; B[i * 3 + j] = acc; }
            }
            if ((carry >> s0) & 1) {
                double *gv = H + 27 + 3 * s0;
                for (int i = 0; i < 3; i++) { double acc = 0; for (int l = 0; l < L; l++) acc += JtW[(size_t)i * L + l] * e->r[l]; gv[i] = acc; }
            }
        }
    }
    if (e) april_graph_factor_eval_destroy(e);
    gp.host_evaluated = nh;
    return chi2;
}
static void upload_host_index(GraphPack &gp) {
    const int nh = (int)gp.host_idx.size();
    if (!nh) return;
    gp.d_host_idx.need(nh); gp.d_hostH.need((size_t)33 * nh);
    HIPCHECK(hipMemcpyAsync(gp.d_host_idx.p, gp.host_idx.data(), (size_t)4 * nh, hipMemcpyHostToDevice, gp.stream));
}

// states (and l_points) of all nodes -> pinned host -> device
static void pack_states(GraphPack &gp, const april_graph_t *g, bool with_lp, bool upload = true) {
    const int N = zsize(g->nodes);
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    if ((size_t)3 * N > gp.h_state.cap || (size_t)3 * N > gp.h_lp.cap || (size_t)3 * N > gp.d_state.cap || (size_t)3 * N > gp.d_lp.cap) gp.mirror_sync = false;   // (a buffer is about to move)
    gp.h_state.need((size_t)3 * N); gp.h_lp.need((size_t)3 * N); gp.h_dx.need((size_t)3 * N);
    for (int i = 0; i < N; i++) {
        if (i + 8 < N) __builtin_prefetch(ns[i + 8]->state);
        const april_graph_node_t *n = ns[i];
        if (n->type != APRIL_GRAPH_NODE_XYT_TYPE || n->length != 3) fail(ERR_UNSUPPORTED, "node %d: only xyt nodes (type 100, 3 DoF) are supported (aprilsam.h:94)", i);
        memcpy(gp.h_state.p + (size_t)3 * i, n->state, 24);
        if (with_lp) memcpy(gp.h_lp.p + (size_t)3 * i, n->l_point, 24);
    }
    gp.N = N;
    gp.pending.clear();           // (every state goes to the device below, or through k_load_states: nothing is left behind the mirrors)
    gp.d_state.need((size_t)3 * N); gp.d_lp.need((size_t)3 * N); gp.d_dx.need((size_t)3 * N);
    if (!upload) return;          // (the batch step reads the pinned mirror from its first kernel, k_load_states)
    HIPCHECK(hipMemcpyAsync(gp.d_state.p, gp.h_state.p, (size_t)24 * N, hipMemcpyHostToDevice, gp.stream));
    if (with_lp) HIPCHECK(hipMemcpyAsync(gp.d_lp.p, gp.h_lp.p, (size_t)24 * N, hipMemcpyHostToDevice, gp.stream));
}

// Incremental steps: compare every node's state / l_point with the pinned mirrors, copy what differs and list those poses
// (gp.changed) -- typically the new pose, the poses the previous step updated, whatever the caller moved.  Returns true when
// patching the listed poses brings the device arrays up to date; false when a full load is needed (mirrors and device not
// known to agree, a buffer had to grow, or too many poses changed for patches to pay).
static long long g_full_reason[4] = { 0 };      // APRILSAM_AMD_INC_PROFILE: why a step loaded every state (mirrors not in step / own updates / caller's changes), steps
static bool pack_states_diff(GraphPack &gp, const april_graph_t *g) {
    const int N = zsize(g->nodes);
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    bool full = !gp.mirror_sync || (size_t)3 * N > gp.d_state.cap || (size_t)3 * N > gp.d_lp.cap || (size_t)3 * N > gp.d_dx.cap;
    gp.h_state.need((size_t)3 * N, true); gp.h_lp.need((size_t)3 * N, true); gp.h_dx.need((size_t)3 * N); gp.h_out.need((size_t)3 * N);
    gp.d_state.need((size_t)3 * N); gp.d_lp.need((size_t)3 * N); gp.d_dx.need((size_t)3 * N);
    gp.changed.clear();
    const int Nold = full ? 0 : gp.N;
    for (int i = 0; i < N; i++) {
        if (i + 8 < N) __builtin_prefetch(ns[i + 8]->state);
        const april_graph_node_t *n = ns[i];
        if (n->type != APRIL_GRAPH_NODE_XYT_TYPE || n->length != 3) fail(ERR_UNSUPPORTED, "node %d: only xyt nodes (type 100, 3 DoF) are supported (aprilsam.h:94)", i);
        double *ms = gp.h_state.p + (size_t)3 * i, *ml = gp.h_lp.p + (size_t)3 * i;
        if (i >= Nold || memcmp(ms, n->state, 24) != 0 || memcmp(ml, n->l_point, 24) != 0) {
            memcpy(ms, n->state, 24); memcpy(ml, n->l_point, 24);
            gp.changed.push_back(i);
        }
    }
    gp.N = N;
    // poses the previous step updated itself (apply_visits brought their mirrors up to date: the walk above found them equal):
    // the device copy of their state is what is stale
    g_full_reason[3]++;
    if (full) g_full_reason[0]++;
    if (!full) {
        if (gp.pending.size() > 48) { full = true; g_full_reason[1]++; }
        else for (int i : gp.pending) if (i < N && std::find(gp.changed.begin(), gp.changed.end(), i) == gp.changed.end()) gp.changed.push_back(i);
    }
    gp.pending.clear();
    if (!full && gp.changed.size() > 48) { full = true; g_full_reason[2]++; }
    return !full;
}

// The same for a step whose walk visits only a few poses (aprilsam.c:755-771, naffected <= 5): what the step READS are the
// l_points / states of the poses of its new factors and of the poses it visits, plus the new poses -- only those are compared
// with the mirrors and patched.  The cost of a step then no longer grows with the size of the graph (the full walk is 1 ns per
// pose per step: 3.5 us on M3500, 100 us on a 100 k-pose graph).  Invariant kept: device arrays == mirrors for EVERY pose;
// mirror == host object only for the poses some call has looked at since -- every consumer that needs all of them (batch
// steps, chi^2, full walks, re-plans) walks all node objects itself.
static bool pack_states_some(GraphPack &gp, const april_graph_t *g, const std::vector<int> &involved) {
    const int N = zsize(g->nodes);
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    if (!gp.mirror_sync || (size_t)3 * N > gp.d_state.cap || (size_t)3 * N > gp.d_lp.cap || (size_t)3 * N > gp.d_dx.cap || (size_t)3 * N > gp.h_state.cap ||
        (size_t)3 * N > gp.h_lp.cap || (size_t)3 * N > gp.h_dx.cap || (size_t)3 * N > gp.h_out.cap || N < gp.N) return pack_states_diff(gp, g);
    gp.changed.clear();
    auto look = [&](int i, bool is_new) {
        const april_graph_node_t *n = ns[i];
        if (n->type != APRIL_GRAPH_NODE_XYT_TYPE || n->length != 3) fail(ERR_UNSUPPORTED, "node %d: only xyt nodes (type 100, 3 DoF) are supported (aprilsam.h:94)", i);
        double *ms = gp.h_state.p + (size_t)3 * i, *ml = gp.h_lp.p + (size_t)3 * i;
        if (is_new || memcmp(ms, n->state, 24) != 0 || memcmp(ml, n->l_point, 24) != 0) {
            memcpy(ms, n->state, 24); memcpy(ml, n->l_point, 24);
            if (std::find(gp.changed.begin(), gp.changed.end(), i) == gp.changed.end()) gp.changed.push_back(i);
        }
    };
    for (int i = gp.N; i < N; i++) look(i, true);
    const int Nold = gp.N;
    for (int i : involved) if (i >= 0 && i < Nold) look(i, false);
    gp.N = N;
    g_full_reason[3]++;
    bool full = false;
    if (gp.pending.size() > 48) { full = true; g_full_reason[1]++; }
    else for (int i : gp.pending) if (i < N && std::find(gp.changed.begin(), gp.changed.end(), i) == gp.changed.end()) gp.changed.push_back(i);
    gp.pending.clear();
    if (!full && gp.changed.size() > 48) { full = true; g_full_reason[2]++; }
    return !full;
}

// evaluation points of the unary factors [from, to): the node's state as packed by this call (april_graph_xytpos.c:83-85
// reads node->state when the factor is evaluated, and the reference evaluates a factor exactly once between batch steps)
static void record_unary_points(GraphPack &gp, int from, int to, const double *states) {
    gp.h_upt.resize((size_t)3 * gp.F, 0.0);
    for (int f = from; f < to; f++) if (gp.h_fb.p[f] < 0) memcpy(&gp.h_upt[(size_t)3 * f], states + (size_t)3 * gp.h_fa.p[f], 24);
}

// ------------------------------------------------------------------------------------------------------
// solver context — one per april_graph_cholesky_param_t pointer
// ------------------------------------------------------------------------------------------------------
enum { K_LINEARIZE = 0, K_FRONT_SMALL, K_ASSEMBLE_BIG, K_DIAG_BIG, K_PANEL_BIG, K_SYRK_BIG, K_BACKSOLVE, K_UPDATE, NKERN };
static const char *const KNAMES[NKERN] = { "k_linearize", "k_front_small", "k_assemble_big", "k_diag_big", "k_panel_big", "k_syrk_big", "k_backsolve", "k_update_states" };
struct Launch { int list_off, pre_off, n, grid; bool single = false; int tile = TILE; };   // single: every front has exactly one work item     // offsets into the int launch-table buffer

struct LevelPlan {
    int small_off = 0, n_small = 0; size_t small_lds = 0;      // fronts handled by k_front_small
    long long full_limit = 0;                                  // ... fully in LDS when their array fits this many bytes, else panel mode
    int small_nt = 512;                                        // ... with this many threads per workgroup
    int n_big = 0; size_t asm_lds = 0;
    Launch asm_big{}, asm_tile{};                              // k_assemble_big (chunks of block columns) / k_assemble_tile (windows, option tile_assembly)
    std::vector<Launch> syrka, syrkb;                          // look-ahead split of the wide update (modes 2, 3), same indexing as syrkw
    std::vector<Launch> panel, syrk, syrkw;                    // per panel step: diag+panel, narrow update, wide update (grid 0 unless the step closes an outer block)
    std::vector<Launch> bchain, btile;                         // per outer block (option block_panels): diagonal-block workgroups, row tiles (Launch::tile = rows per wave / 16)
    std::vector<int> diag_slot0;                               // per panel step: first slot of its factored diagonal blocks in d_diag (multi-tile steps)
    int wb_off = 0, n_wb = 0, n_diag_slots = 0;                // k_diag_writeback entries (3 ints each) of the level, slots used
    int all_off = 0, n_all = 0; size_t solve_lds = 0;          // every front (k_backsolve)
    size_t solve_w_lds = 0; int maxns = 0;                     // ... in the column-per-lane form (k_backsolve_w: L panel in LDS), widest own part
    Launch bs_gemv{};                                          // fronts whose update-row product is spread over workgroups first (k_backsolve_gemv)
    Launch bs_blk{}; size_t bs_blk_lds = 0;                    // wide fronts back-substituted by a chain + helper workgroups (k_backsolve_blk); grid 0: none
    int rest_off = 0, n_rest = 0; size_t rest_lds = 0;         // ... and the level's other fronts (k_backsolve_t)
};

// state of the incremental fast path (inc_fast.*): the plan of the last batch step stays frozen, poses added since
// form one growing TAIL front at the root, and only fronts on the root paths of changed leaves are regenerated
struct IncState {
    bool ready = false;                     // helper tables below are built for the current base plan
    int Nb = 0, Fb = 0, nF0 = 0, nLev0 = 0;
    int cap_nodes = 0, cap_fact = 0;         // slack reserved at plan upload (0 until the param has seen an incremental call)
    long long i32_used = 0, dest_used = 0, child_used = 0, tab_used = 0, pool_used = 0, pool_cap = 0, o_rows = 0, o_rel = 0;
    int slots_used = 0;
    std::vector<FrontDesc> fd;              // host mirror of the device descriptors (base fronts, then TAIL)
    std::vector<int> pos_front;             // base position -> base front
    std::vector<int> parent;                // current assembly parent (base roots get TAIL once they see tail rows)
    std::vector<std::vector<int>> E;        // per base front: tail nodes in its extended struct (sorted)
    std::vector<std::vector<int>> xfac;     // per front (TAIL = index nF0): factors added since the batch
    std::vector<int> bf_ptr, bf_idx;        // base factors owned by each base front (CSR)
    std::vector<int> rel_begin;             // per front: absolute offset of its current child->parent block map
    std::vector<int> cur_nub;               // per front: current update blocks
    std::vector<long long> cur_cap;         // per front: doubles allocated at fd.off
    std::vector<char> dirty;
    std::vector<int> f_level;               // base levels, tail front i = nLev0 + i
    int zpos = 0;                           // a position whose x entries stay zero: what the phantom rows of the last tail front point at
    int tail_ok = -1;                       // tail front whose factor on the device is complete in the padded layout (candidate for tail_refactor), or -1
    int recs_stale = -1;                    // tail front whose destination records on the device lack the factors tail_refactor took in directly, or -1
    std::vector<int> t_first, t_cnt;        // tail fronts: first pose id, own poses
    std::vector<int> tf_of;                 // tail pose (id - Nb) -> its tail front
    std::vector<std::vector<int>> kids;     // per front: children that are NOT in the base plan's child lists (children of tail fronts)
    std::vector<LevelPlan> base_levels;     // launch tables of all fronts per level for the back substitution
    // staging for the per-step uploads (members: the async copies read them until the step's final sync)
    std::vector<int> st_i32, st_tab, st_sb, st_sr, st_ids; std::vector<DestRec> st_dest; std::vector<ChildRec> st_child;
    std::vector<unsigned char> st_sw; std::vector<double> st_zeros; std::vector<char> need;
    // low-rank updates of the fronts on a loop closure's root path (front_update_body, option inc_update)
    std::vector<char> stale;                // per front: its destination / child records on the device lack what update steps folded in directly
    bool upd_ok = true;                     // false after a step that failed half-way, until the next full plan
    bool base_has_big = false;              // the base plan has fronts on the multi-workgroup path
    std::vector<char> st_mid, st_mode; std::vector<int> st_owner, st_mask, st_wout, st_slot; std::vector<UpdRec> st_upd, st_rec;
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
    DBuf<double> d_pool, d_H, d_x, d_diag;   // d_diag: factored diagonal blocks of the current panel step, one per active big front
    DBuf<int> d_bad;
    HBuf<int> h_bad;
    IncPrologue pro{}; InlinePatches inl{};       // arguments of the incremental step's first kernel
    bool no_speculation = false;          // batch_impl: the next call reads the factor objects before it launches (set when a speculative run was voided)
    HBuf<long long> h_kstamp; HBuf<int> h_done; int done_seq = 0, one_wait = 0;      // k_inc_one: phase stamps (profile), completion word the host spins on
    std::vector<double> h_lambda;
    aprilsam_amd_stats_t st{};
    hipEvent_t ev[8] = {};
    bool have_events = false;
    // per-kernel HIP-event timing (instrumented passes only)
    double k_ms[NKERN] = {}; long long k_calls[NKERN] = {};
    std::vector<hipEvent_t> k_ev; std::vector<int> k_ids;
    // incremental bookkeeping (aprilsam.c:741-751, 566-575)
    bool have_fact = false;               // a batch factorisation exists (reference: param->chol != NULL)
    int plan_pin = 0;                     // pin_last the plan was built with
    long long plan_persist = 0;           // launch_table_key() the launch tables were built with
    int same_topo_batches = 0;            // batch calls on an extended (base + tail fronts) plan whose topology did not change since the previous call
    bool used_inc = false;                // april_graph_cholesky_inc has been called on this param
    bool want_inc = false;                // the param has been used incrementally: plan uploads reserve the append slack
    int batch_nodes = 0;                  // #nodes at the last batch step (those carry the Tikhonov term)
    IncState inc;
    int inc_F = 0, inc_N = 0;                      // factors / nodes folded into the factorisation so far
    std::vector<RefModel::Visit> visits; std::vector<int> involved;
    std::vector<int> base_tab;                     // host copy of the launch tables of the base plan
    std::vector<int> inc_slot_blk, inc_slot_rhs;   // slots of the factors added since the base plan (3 / 2 per factor)
    RefModel model;                       // the reference's tree / counters (refmodel.cpp), rebuilt lazily after a batch
    PatchList patches;                    // per-step table updates of the incremental fast path
    int batch_factors = 0;                // #factors at the last batch step
    // look-ahead: the "rest" part of the wide trailing updates runs on a side stream (enqueue_big_steps)
    hipStream_t s2 = nullptr; std::vector<hipEvent_t> la_ev; size_t la_next = 0;
    hipEvent_t la_event() {
        if (la_next == la_ev.size()) { hipEvent_t e; HIPCHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); la_ev.push_back(e); }
        return la_ev[la_next++];
    }
    // captured numeric phase
    // multi-level ("persistent") launches of the batch path: the top levels of the tree, where a level holds only a handful
    // of fronts, run as ONE launch for the factorisation and ONE for the back substitution, fronts waiting on per-front
    // dependency flags instead of on kernel boundaries (kernels.hip.h: wait_flag / publish_flag)
    int persist_l0 = -1;                  // first level of the multi-level launch, -1: none
    int p_up_off = 0, p_up_n = 0, p_dn_off = 0, p_dn_n = 0, p_nt = 1024; size_t p_up_lds = 0, p_dn_lds = 0; long long p_up_full = 0; int p_dn_maxns = 0;
    DBuf<int> d_flags, d_flevel, d_perm;
    DBuf<double> d_dinv, d_bsb_far; DBuf<int> d_bsb_flags;   // inverse diagonal blocks of the big fronts; scratch of k_backsolve_blk
    DBuf<int> d_solve_tab; std::vector<int> solve_tab;      // april_graph_cholesky_inc_solver: front lists of its back substitution
    DBuf<UpdRec> d_upd; DBuf<double> d_wbuf;               // incremental steps: update records per launch-list entry, the step's travelling vectors
    hipGraphExec_t gexec = nullptr;
    const void *gexec_key = nullptr;      // GraphPack the graph was captured against
    long long gexec_serial = 0;
    // the same phase as the API call runs it: first kernel reads the caller's states from the pinned mirror, last kernel
    // writes new states / dx / pivot flag back to pinned mirrors -- one graph launch + one stream sync per call
    hipGraphExec_t gexec_api = nullptr;
    const void *api_key[7] = {};
    double lambda_val = -1; int lambda_N = -1;     // what d_lambda currently holds (uniform batch value), -1: unknown
    void release() {
        d_i32.release(); d_fd.release(); d_dest.release(); d_child.release(); d_lambda.release(); d_tab.release(); d_swap.release(); d_pos.release();
        d_pool.release(); d_H.release(); d_x.release(); d_diag.release(); d_bad.release(); h_bad.release(); patches.release();
        h_done.release(); h_kstamp.release(); d_prof.release(); d_upd.release(); d_wbuf.release(); d_flags.release(); d_flevel.release(); d_perm.release(); d_solve_tab.release(); d_dinv.release(); d_bsb_far.release(); d_bsb_flags.release();
        if (gexec) (void)hipGraphExecDestroy(gexec);
        gexec = nullptr;
        if (gexec_api) (void)hipGraphExecDestroy(gexec_api);
        gexec_api = nullptr;
        if (have_events) for (auto &e : ev) (void)hipEventDestroy(e);
        have_events = false;
        for (auto &e : k_ev) (void)hipEventDestroy(e);
        k_ev.clear();
        for (auto &e : la_ev) (void)hipEventDestroy(e);
        la_ev.clear(); la_next = 0;
        if (s2) (void)hipStreamDestroy(s2);
        s2 = nullptr;
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

// ------------------------------------------------------------------------------------------------------
// failure path (errors.h): every entry point below runs its body inside guarded().  A SolverError thrown anywhere under
// it unwinds to here (the body's lock_guard is released on the way): message on stderr, code kept for
// aprilsam_amd_last_error / stats.error_code, and the solver state that may be half-built -- the graph's pack, the param's
// plan, fronts, captured graphs, sharding state -- is dropped wholesale, exactly as param_destory / graph_destroy would.
// The caller's node objects are only ever written after a call's final stream sync succeeded, so they are untouched.
// ------------------------------------------------------------------------------------------------------
struct ShardState;
static void drop_shard_state(const void *param);
static void on_failure(const april_graph_cholesky_param_t *param, const april_graph_t *g, int code, const std::string &msg) {
    set_last_error(code, msg);
    fprintf(stderr, "aprilsam_amd: ERROR %d: %s -- node states left untouched\n", code, msg.c_str());
    fflush(stderr);
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipGetLastError();
    if (g) {
        auto it = g_packs.find(g);
        if (it != g_packs.end()) {
            hipStream_t s = it->second->stream;
            if (s) {
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {      // failed inside a capture
                    hipGraph_t gr = nullptr;
                    (void)hipStreamEndCapture(s, &gr);
                    if (gr) (void)hipGraphDestroy(gr);
                }
                (void)hipStreamSynchronize(s);           // whatever was enqueued before the failure
                (void)hipGetLastError();
            }
            it->second->release(); g_packs.erase(it);
        }
    }
    if (param) {
        drop_shard_state(param);
        auto it = g_ctx.find(param);
        aprilsam_amd_stats_t st{};
        if (it != g_ctx.end()) { st = it->second->st; it->second->release(); }
        auto fresh = std::make_unique<Context>();
        fresh->st = st; fresh->st.error_code = code; fresh->st.not_spd = 0;
        g_ctx[param] = std::move(fresh);
    }
}
template <class Fn> static void guarded(const april_graph_cholesky_param_t *param, const april_graph_t *g, Fn &&fn) {
    try { fn(); }
    catch (const SolverError &e) { on_failure(param, g, e.code, e.msg); }
    catch (const std::bad_alloc &) { on_failure(param, g, ERR_OOM, "host memory exhausted (std::bad_alloc)"); }
    catch (const std::exception &e) { on_failure(param, g, ERR_INTERNAL, e.what()); }
}
template <class Fn> static int guarded_rc(const april_graph_cholesky_param_t *param, const april_graph_t *g, Fn &&fn) {
    try { return fn(); }
    catch (const SolverError &e) { on_failure(param, g, e.code, e.msg); return e.code; }
    catch (const std::bad_alloc &) { on_failure(param, g, ERR_OOM, "host memory exhausted (std::bad_alloc)"); return ERR_OOM; }
    catch (const std::exception &e) { on_failure(param, g, ERR_INTERNAL, e.what()); return ERR_INTERNAL; }
}

// slack reserved at plan upload so that the incremental path can append without reallocating device buffers
constexpr int INC_NODES = 4096, INC_FACT = 16384, INC_I32 = 4 << 20, INC_DEST = 1 << 20, INC_CHILD = 1 << 18, INC_TAB = 1 << 20;
#define TAIL_POSES (g_opt.tail_poses)         // own poses per tail front of the incremental path (inc_fast_step), option tail_poses (>= 8)
constexpr int MAX_TAIL_FRONTS = INC_NODES / 8 + 8;
constexpr long long INC_POOL_MIN = 8ll << 20;             // doubles (64 MB; the M3500 demo appends ~25 MB of regenerated fronts between two batch steps)

// waves of a k_front_small workgroup (option small_threads)
static int waves_of(int nt) { return nt >= 1024 ? 16 : (nt >= 512 ? 8 : 4); }
// workgroup size of k_front_small on a level with n fronts: latency levels take the big workgroup (more lanes on one
// front's critical path), throughput levels the smaller one (more workgroups per CU)
static long long g_incfail[32] = { 0 };        // APRILSAM_AMD_INC_PROFILE: why inc_fast_step handed a step to a full re-plan (exit number in source order)
static bool inc_fail(int why) { g_incfail[why & 31]++; return false; }
static long long g_updstat[6] = { 0 };      // APRILSAM_AMD_INC_PROFILE: general-path steps with / without updated fronts, fronts updated / re-factorised in the former, re-factorised in the latter, steps through k_inc_one
static double g_incsub[8] = { 0 }; static long long g_incsub_n = 0;      // APRILSAM_AMD_INC_PROFILE: host sub-phases of the general incremental path (ms, summed)
static const bool g_incprof_stamps = [] { const char *e = getenv("APRILSAM_AMD_INC_PROFILE"); return e && *e == '2'; }();      // (see IncProf)
static int small_threads_for(size_t n_fronts) { return (int)n_fronts >= g_opt.tp_fronts ? std::min(g_opt.small_threads, g_opt.tp_threads) : g_opt.small_threads; }

// doubles of d_diag a level needs: one (NB x NB+1) slot per parked diagonal block (per-panel forms) or four NB x NB inverse
// blocks per active front (outer-block panels)
static size_t diag_doubles(int n_big, int n_diag_slots) {
    return std::max((size_t)(std::max(n_big, n_diag_slots) + 64) * NB * (NB + 1), (size_t)std::max(n_big, 1) * OBP * NB * NB);
}
// classify the fronts of one level (small / big) and append their launch tables to `tab`
constexpr int BSB_MAX_WGS = 64;               // chain + helper workgroups of one k_backsolve_blk launch (they must be resident together)
template <class Dims, class KeepInv>
static void build_level(LevelPlan &L, std::vector<int> &fronts, std::vector<int> &tab, Dims dims, KeepInv keep_inv) {
    const size_t small_max = (size_t)g_opt.small_lds_kb * 1024;
    L = LevelPlan();
    std::vector<int> small, big;
    size_t maxm = 0;
    auto rows = [&](int t) { int a, b; dims(t, &a, &b); return 3 * (a + b + 1); };
    auto cols = [&](int t) { int a, b; dims(t, &a, &b); return 3 * (a + b); };
    auto nsb_of = [&](int t) { int a, b; dims(t, &a, &b); return a; };
    // throughput levels (far more fronts than compute units): a lower full-LDS limit sends mid-size fronts to panel mode,
    // whose LDS footprint (own columns only) lets several workgroups share a CU
    size_t full_max = small_max;
    if ((int)fronts.size() >= g_opt.tp_fronts && g_opt.tp_lds_kb > 0) full_max = std::min(small_max, (size_t)g_opt.tp_lds_kb * 1024);
    L.full_limit = (long long)full_max;
    L.small_nt = small_threads_for(fronts.size());
    const int nw = waves_of(L.small_nt);
    for (int t : fronts) {
        const int R = rows(t), C = cols(t);
        maxm = std::max<size_t>(maxm, C);
        L.solve_w_lds = std::max(L.solve_w_lds, backsolve_lds(C, 3 * nsb_of(t), true)); L.maxns = std::max(L.maxns, 3 * nsb_of(t));
        const size_t lds_s = small_front_lds(R, C, nw), lds_p = panel_front_lds(R, 3 * nsb_of(t), nw);
        if (lds_s <= full_max) { small.push_back(t); L.small_lds = std::max(L.small_lds, lds_s); }
        else if (g_opt.panel_mode && lds_p <= small_max) { small.push_back(t); L.small_lds = std::max(L.small_lds, lds_p); }   // k_front_small, panel mode
        else if (lds_s <= small_max) { small.push_back(t); L.small_lds = std::max(L.small_lds, lds_s); L.full_limit = std::max(L.full_limit, (long long)lds_s); }
        else big.push_back(t);
    }
    // longest-processing-time first: the widest fronts of a level start first
    std::sort(small.begin(), small.end(), [&](int a, int b) { int ra = rows(a), rb = rows(b); return ra != rb ? ra > rb : a < b; });
    L.all_off = (int)tab.size(); L.n_all = (int)fronts.size();
    tab.insert(tab.end(), fronts.begin(), fronts.end());
    {   // back substitution: fronts with a large update block get their update-row product from k_backsolve_gemv
        std::vector<int> sp;
        for (int t : fronts) { int a, b; dims(t, &a, &b); if (bs_split_front(a, b)) sp.push_back(t); }
        L.bs_gemv = Launch{ (int)tab.size(), 0, (int)sp.size(), 0, false };
        tab.insert(tab.end(), sp.begin(), sp.end());
        L.bs_gemv.pre_off = (int)tab.size();
        int acc = 0; tab.push_back(0);
        for (int t : sp) { acc += (3 * nsb_of(t) + NB - 1) / NB; tab.push_back(acc); }
        L.bs_gemv.grid = acc;
    }
    L.solve_lds = (maxm + NB + 8 + NB * (NB + 1)) * 8;
    L.small_off = (int)tab.size(); L.n_small = (int)small.size();
    tab.insert(tab.end(), small.begin(), small.end());
    L.n_big = (int)big.size();
    if (big.empty()) return;
    std::sort(big.begin(), big.end(), [&](int a, int b) { return nsb_of(a) != nsb_of(b) ? nsb_of(a) > nsb_of(b) : a < b; });
    int list_off = (int)tab.size();
    tab.insert(tab.end(), big.begin(), big.end());
    auto make = [&](int nact, auto count) {
        Launch La; La.list_off = list_off; La.n = nact; La.pre_off = (int)tab.size();
        int acc = 0; tab.push_back(0);
        for (int i = 0; i < nact; i++) { acc += count(big[i]); tab.push_back(acc); }
        La.grid = acc; La.single = acc == nact;
        return La;
    };
    L.asm_big = make((int)big.size(), [&](int t) { return asm_chunks(cols(t) / 3); });
    L.asm_tile = make((int)big.size(), [&](int t) { return at_tiles(cols(t) / 3); });
    {   // back substitution of the multi-workgroup fronts 128 columns at a time (k_backsolve_blk): fronts whose inverse
        // diagonal blocks are kept (keep_inv) and whose update-row product comes from k_backsolve_gemv (or is empty);
        // the whole level or nothing: chains + helpers must fit BSB_MAX_WGS workgroups
        std::vector<int> wide, rest, bigs(big);
        std::sort(bigs.begin(), bigs.end());
        int wgs = 0;
        for (int t : fronts) {
            int a, b; dims(t, &a, &b);
            const bool isbig = std::binary_search(bigs.begin(), bigs.end(), t);
            if (isbig && 3 * a <= BSB_FAR && (b == 0 || bs_split_front(a, b)) && keep_inv(t)) { wide.push_back(t); wgs += 1 + bsb_helpers(3 * a); }
            else rest.push_back(t);
        }
        if (!wide.empty() && wgs <= BSB_MAX_WGS && g_opt.block_panels && g_opt.blk_backsolve) {
            const int lo = (int)tab.size();
            tab.insert(tab.end(), wide.begin(), wide.end());
            L.bs_blk = Launch{ lo, (int)tab.size(), (int)wide.size(), 0, false };
            int acc = 0; tab.push_back(0);
            for (int t : wide) { acc += 1 + bsb_helpers(3 * nsb_of(t)); tab.push_back(acc); L.bs_blk_lds = std::max(L.bs_blk_lds, bsb_lds(3 * nsb_of(t))); }
            L.bs_blk.grid = acc;
            L.rest_off = (int)tab.size(); L.n_rest = (int)rest.size();
            tab.insert(tab.end(), rest.begin(), rest.end());
            size_t mm = 0; for (int t : rest) mm = std::max<size_t>(mm, cols(t));
            L.rest_lds = (mm + NB + 8 + NB * (NB + 1)) * 8;
        }
    }
    int steps = (3 * nsb_of(big[0]) + NB - 1) / NB;
    auto active = [&](int sidx) { int nact = 0; while (nact < (int)big.size() && 3 * nsb_of(big[nact]) > sidx * NB) nact++; return nact; };
    std::vector<int> wb;
    for (int sidx = 0; sidx < steps; sidx++) {
        const int nact = active(sidx);
        L.panel.push_back(make(nact, [&](int t) { return panel_tiles(rows(t), 3 * nsb_of(t), sidx, g_opt.left_panels && g_opt.fused_panel ? PANEL_ROWS_LL : PANEL_ROWS); }));
        L.diag_slot0.push_back(L.n_diag_slots);
        if (!L.panel.back().single || (g_opt.left_panels && g_opt.fused_panel)) {   // the factored diagonal blocks wait in d_diag until the level's write-back
            for (int i = 0; i < nact; i++) { wb.push_back(big[i]); wb.push_back(sidx); wb.push_back(L.n_diag_slots + i); }
            L.n_diag_slots += nact;
        }
        L.syrk.push_back(make(nact, [&](int t) { return syrk_tiles(rows(t), cols(t), 3 * nsb_of(t), sidx, sidx + 1, 0); }));
        if ((sidx + 1) % OBP == 0 || sidx + 1 == steps) {
            const int s_lo = sidx / OBP * OBP;
            // wide trailing matrices go to the LDS-staged 128 x 128 kernel (decided per launch on the largest front)
            int span = 0;
            for (int i = 0; i < active(s_lo); i++) { SyrkRange r = syrk_range(rows(big[i]), cols(big[i]), 3 * nsb_of(big[i]), s_lo, sidx + 1, 1); if (r.ntr > 0) span = std::max(span, rows(big[i]) - 2 - r.col_lo); }
            const int tile = span >= g_opt.syrk128_rows ? TILE2 : TILE;
            L.syrkw.push_back(make(active(s_lo), [&](int t) { return syrk_tiles(rows(t), cols(t), 3 * nsb_of(t), s_lo, sidx + 1, 1, tile); }));
            L.syrkw.back().tile = tile;
            L.syrka.push_back(make(active(s_lo), [&](int t) { return syrk_tiles(rows(t), cols(t), 3 * nsb_of(t), s_lo, sidx + 1, 2); }));
            L.syrkb.push_back(make(active(s_lo), [&](int t) { return syrk_tiles(rows(t), cols(t), 3 * nsb_of(t), s_lo, sidx + 1, 3, tile); }));
            L.syrkb.back().tile = tile;
        } else {
            L.syrkw.push_back(Launch{ 0, 0, 0, 0, false });
            L.syrka.push_back(Launch{ 0, 0, 0, 0, false }); L.syrkb.push_back(Launch{ 0, 0, 0, 0, false });
        }
    }
    L.wb_off = (int)tab.size(); L.n_wb = (int)wb.size() / 3;
    tab.insert(tab.end(), wb.begin(), wb.end());
    // outer-block panels (kernels.hip.h: k_block_chain / k_block_solve): per 128-column outer block the active fronts (a
    // prefix of `big`, sorted by own columns) and their row tiles below the diagonal block
    for (int o = 0; o * OBP < steps; o++) {
        const int nact = active(o * OBP);
        Launch bc{ list_off, 0, nact, nact, true };
        L.bchain.push_back(bc);
        long long tiles1 = 0;
        for (int i = 0; i < nact; i++) tiles1 += block_tiles(rows(big[i]), 3 * nsb_of(big[i]), o, 1);
        const int rb = tiles1 > 1024 ? 2 : 1;          // many tiles: 32 rows per wave (half the workgroups, each staging the same L11 once)
        L.btile.push_back(make(nact, [&](int t) { return block_tiles(rows(t), 3 * nsb_of(t), o, rb); }));
        L.btile.back().tile = rb;
    }
}

// Per-rank layout of the front pool in a sharded run: a rank keeps the frontal arrays of the fronts it OWNS and, for
// every child of an owned front that lives on another rank, a "ghost" holding only that child's update block
// ((3 cnu + 3) rows x 3 cnu columns: what the parent's extend-add reads, filled from the wire).  off < 0: not present.
struct ShardLayout { std::vector<long long> off; std::vector<char> ghost; long long pool_doubles = 0; };

// upload the symbolic plan and build the per-level launch tables
static void upload_plan(Context &c, hipStream_t s, const ShardLayout *lay = nullptr) {
    const Plan &P = c.plan;
    if (c.gexec) { (void)hipGraphExecDestroy(c.gexec); c.gexec = nullptr; }
    if (c.gexec_api) { (void)hipGraphExecDestroy(c.gexec_api); c.gexec_api = nullptr; }
    c.lambda_N = -1;
    // ---- descriptors + index arrays ----------------------------------------------------------------------------
    std::vector<FrontDesc> &fd = c.inc.fd; fd.assign(P.nF, FrontDesc());
    std::vector<ChildRec> ch(std::max<size_t>(1, P.ch_idx.size()));
    for (int t = 0; t < P.nF; t++) {
        FrontDesc &d = fd[t];
        memset(&d, 0, sizeof(d));
        d.off = lay ? std::max<long long>(lay->off[t], 0) : P.f_off[t]; d.nsb = P.f_nsb[t]; d.nub = P.f_nub[t]; d.first = P.f_first[t];
        d.dest_begin = P.dest_front_ptr[t]; d.dest_end = P.dest_front_ptr[t + 1];
        d.ch_begin = P.ch_ptr[t]; d.ch_end = P.ch_ptr[t + 1];
        d.rows_begin = 0; d.parent = P.f_parent[t];      // rows_begin patched below (absolute offset in the int arena)
    }
    for (size_t k = 0; k < P.ch_idx.size(); k++) {
        const int cfr = P.ch_idx[k];
        ChildRec &r = ch[k];
        r.cR = P.rows(cfr); r.cnu = P.f_nub[cfr];
        r.uoff = P.f_off[cfr] + (long long)(3 * P.f_nsb[cfr]) * r.cR + 3 * P.f_nsb[cfr];
        if (lay) {
            if (lay->ghost[cfr]) { r.cR = 3 * r.cnu + 3; r.uoff = lay->off[cfr]; }
            else r.uoff = std::max<long long>(lay->off[cfr], 0) + (long long)(3 * P.f_nsb[cfr]) * r.cR + 3 * P.f_nsb[cfr];
        }
        r.rel_begin = 0; r.pad = cfr;                  // rel_begin patched below; pad keeps the child's front id
    }
    // slack for the incremental path is only reserved once the param has been used incrementally (a 3-node tutorial graph
    // solved in batch mode should not cost hundreds of MB of HBM); the first incremental call then re-plans once
    const bool inc = c.want_inc;
    const int INC_NODES_ = inc ? INC_NODES : 0, INC_FACT_ = inc ? INC_FACT : 0;
    const size_t INC_I32_ = inc ? INC_I32 : 0, INC_DEST_ = inc ? INC_DEST : 0, INC_CHILD_ = inc ? INC_CHILD : 0, INC_TAB_ = inc ? INC_TAB : 0;
    c.inc.cap_nodes = INC_NODES_; c.inc.cap_fact = INC_FACT_;
    std::vector<int> i32;
    auto put32 = [&](const std::vector<int> &v) { size_t o = i32.size(); i32.insert(i32.end(), v.begin(), v.end()); if (v.empty()) i32.push_back(0); return o; };
    size_t o_rows = put32(P.f_rows), o_rel = put32(P.f_rel);
    size_t o_sb = put32(P.slot_blk); i32.resize(i32.size() + (size_t)3 * INC_FACT_, -1);      // room for factors added incrementally
    size_t o_sr = put32(P.slot_rhs); i32.resize(i32.size() + (size_t)2 * INC_FACT_, -1);
    c.inc.i32_used = (long long)i32.size(); c.inc.dest_used = (long long)P.dest.size(); c.inc.child_used = (long long)P.ch_idx.size();
    for (int t = 0; t < P.nF; t++) fd[t].rows_begin = (int)(o_rows + P.f_rows_ptr[t]);
    for (size_t k = 0; k < P.ch_idx.size(); k++) ch[k].rel_begin = (int)(o_rel + P.f_rows_ptr[ch[k].pad]);
    c.d_i32.need(i32.size() + INC_I32_); c.d_fd.need(fd.size() + 1 + (inc ? MAX_TAIL_FRONTS : 0)); c.d_child.need(ch.size() + INC_CHILD_);
    c.d_dest.need(std::max<size_t>(1, P.dest.size()) + INC_DEST_);
    HIPCHECK(hipMemcpyAsync(c.d_i32.p, i32.data(), i32.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(c.d_fd.p, fd.data(), fd.size() * sizeof(FrontDesc), hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(c.d_child.p, ch.data(), ch.size() * sizeof(ChildRec), hipMemcpyHostToDevice, s));
    static_assert(sizeof(DestRec) == sizeof(Plan::DestRec), "DestRec layout");
    if (!P.dest.empty()) HIPCHECK(hipMemcpyAsync(c.d_dest.p, P.dest.data(), P.dest.size() * sizeof(DestRec), hipMemcpyHostToDevice, s));
    c.d_lambda.need((size_t)P.N + INC_NODES_);
    DevPlan &d = c.dp;
    d.nF = P.nF;
    d.fd = c.d_fd.p; d.dest = c.d_dest.p; d.child = c.d_child.p;
    d.f_rows = c.d_i32.p; d.f_rel = c.d_i32.p; d.slot_blk = c.d_i32.p + o_sb; d.slot_rhs = c.d_i32.p + o_sr; d.src_idx = c.d_i32.p;
    c.inc.o_rows = (long long)o_rows; c.inc.o_rel = (long long)o_rel;
    d.lambda = c.d_lambda.p;
    d.prof = nullptr; d.prof_mode = 0;
    if (getenv("APRILSAM_AMD_KPROF")) { c.d_prof.need((size_t)PROF_SLOTS * P.nF); HIPCHECK(hipMemsetAsync(c.d_prof.p, 0, (size_t)8 * PROF_SLOTS * P.nF, s)); d.prof = c.d_prof.p; d.prof_mode = atoi(getenv("APRILSAM_AMD_KPROF")) >= 2 ? atoi(getenv("APRILSAM_AMD_KPROF")) : 1; }
    c.d_swap.need((size_t)P.F + INC_FACT_); c.d_pos.need((size_t)P.N + INC_NODES_);
    HIPCHECK(hipMemcpyAsync(c.d_swap.p, P.fac_swap.data(), P.F, hipMemcpyHostToDevice, s));
    HIPCHECK(hipMemcpyAsync(c.d_pos.p, P.pos.data(), (size_t)P.N * 4, hipMemcpyHostToDevice, s));
    c.d_perm.need((size_t)P.N + INC_NODES_);
    HIPCHECK(hipMemcpyAsync(c.d_perm.p, P.perm.data(), (size_t)P.N * 4, hipMemcpyHostToDevice, s));

    // ---- launch tables -------------------------------------------------------------------------------------
    std::vector<int> tab;
    c.levels.assign(P.nLevels, LevelPlan());
    for (int l = 0; l < P.nLevels; l++) {
        std::vector<int> fr(P.lev_fronts.begin() + P.lev_ptr[l], P.lev_fronts.begin() + P.lev_ptr[l + 1]);
        build_level(c.levels[l], fr, tab, [&](int t, int *nsb, int *nub) { *nsb = P.f_nsb[t]; *nub = P.f_nub[t]; }, [](int) { return true; });
    }
    {   // persistent inverse diagonal blocks of the multi-workgroup fronts (k_block_chain leaves them, k_block_solve and
        // k_backsolve_blk use them): whole outer blocks per front
        long long slots = 0;
        for (int t = 0; t < P.nF; t++) fd[t].dinv0 = -1;
        for (int l = 0; l < P.nLevels; l++) {
            const LevelPlan &L = c.levels[l];
            for (int k = 0; k < L.asm_big.n; k++) { const int t = tab[L.asm_big.list_off + k]; fd[t].dinv0 = (int)slots; slots += (long long)OBP * bsb_blocks(3 * P.f_nsb[t]); }
        }
        c.d_dinv.need((size_t)std::max<long long>(slots, 1) * NB * NB);
        HIPCHECK(hipMemcpyAsync(c.d_fd.p, fd.data(), fd.size() * sizeof(FrontDesc), hipMemcpyHostToDevice, s));
        if (!c.d_bsb_flags.p) { c.d_bsb_flags.need((size_t)BSB_MAX_WGS * 2 * BSB_MAXB); HIPCHECK(hipMemsetAsync(c.d_bsb_flags.p, 0, c.d_bsb_flags.cap * 4, s)); c.d_bsb_far.need((size_t)BSB_MAX_WGS * BSB_FAR); }
    }
    for (int l = 0; l < P.nLevels; l++)
        if (c.levels[l].solve_lds > 160 * 1024)       // k_backsolve keeps x over a front's rows in LDS (~19 000 scalar rows)
            fail(ERR_UNSUPPORTED, "a frontal matrix has more rows than the back-substitution kernel can hold in LDS (an unsplittable dense region "
                 "of more than ~6000 poses); this build does not tile the solve of such a front");
    // ---- multi-level launch over the top of the tree (small fronts only, few per level) ------------------------------
    c.persist_l0 = -1;
    if (g_opt.persist && !lay && P.nLevels >= 3) {
        int l0 = P.nLevels, cnt = 0;
        const int nt_top = c.levels[P.nLevels - 1].small_nt;
        for (int l = P.nLevels - 1; l >= 0; l--) {           // (level 0 too when persist_max_fronts allows: small graphs run as one launch per sweep)
            const LevelPlan &L = c.levels[l];
            if (L.n_big > 0 || L.bs_gemv.grid > 0 || L.small_nt != nt_top || L.n_small != L.n_all || cnt + L.n_small > g_opt.persist_max_fronts) break;
            cnt += L.n_small; l0 = l;
        }
        if (P.nLevels - l0 >= 2) {
            c.persist_l0 = l0; c.p_nt = nt_top; c.p_up_lds = 0; c.p_dn_lds = 0; c.p_up_full = 0; c.p_dn_maxns = 0;
            c.p_up_off = (int)tab.size(); c.p_up_n = cnt;
            for (int l = l0; l < P.nLevels; l++) {                     // children before parents: dependencies have lower workgroup ids
                const LevelPlan &L = c.levels[l];
                for (int k = 0; k < L.n_small; k++) tab.push_back(tab[L.small_off + k]);
                c.p_up_lds = std::max(c.p_up_lds, L.small_lds); c.p_up_full = std::max(c.p_up_full, L.full_limit);
                for (int k = 0; k < L.n_all; k++) { const int t = tab[L.all_off + k]; c.p_dn_lds = std::max(c.p_dn_lds, backsolve_lds(P.cols(t), 3 * P.f_nsb[t], true)); c.p_dn_maxns = std::max(c.p_dn_maxns, 3 * P.f_nsb[t]); }
            }
            c.p_dn_off = (int)tab.size(); c.p_dn_n = cnt;
            for (int l = P.nLevels - 1; l >= l0; l--) { const LevelPlan &L = c.levels[l]; for (int k = 0; k < L.n_all; k++) tab.push_back(tab[L.all_off + k]); }
        }
    }
    // (dependency flags / front levels: also used by the extended-plan batch step, whose tail fronts get levels of their own)
    c.d_flags.need((size_t)3 * (P.nF + MAX_TAIL_FRONTS)); c.d_flevel.need((size_t)P.nF + MAX_TAIL_FRONTS);      // done / x done / vectors ready
    if (inc) { c.d_upd.need((size_t)g_opt.persist_max_fronts + 64); c.d_wbuf.need((size_t)1 << 19); }
    HIPCHECK(hipMemcpyAsync(c.d_flevel.p, P.f_level.data(), (size_t)P.nF * 4, hipMemcpyHostToDevice, s));
    if (tab.empty()) tab.push_back(0);
    c.d_tab.need(tab.size() + INC_TAB_);
    c.inc.tab_used = (long long)tab.size();
    c.base_tab = tab;
    HIPCHECK(hipMemcpyAsync(c.d_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHECK(hipStreamSynchronize(s));      // host vectors above go out of scope

    const long long pool_slack = inc ? std::max<long long>(INC_POOL_MIN, P.pool_doubles / 4) : 0;
    const long long pool_doubles = lay ? lay->pool_doubles : P.pool_doubles;
    c.d_pool.need((size_t)std::max<long long>(pool_doubles, 1) + (size_t)pool_slack);
    c.inc.pool_used = pool_doubles; c.inc.pool_cap = (long long)c.d_pool.cap;
    c.d_H.need((size_t)9 * ((size_t)std::max(1, P.n_slots) + (size_t)5 * INC_FACT_)); c.d_x.need((size_t)3 * ((size_t)P.N + INC_NODES_ + 1));
    c.inc.zpos = P.N + INC_NODES_;
    HIPCHECK(hipMemsetAsync(c.d_x.p + (size_t)3 * c.inc.zpos, 0, 24, s));
    c.inc.slots_used = P.n_slots;
    c.inc.ready = false; c.inc.t_first.clear(); c.same_topo_batches = 0;
    c.d_bad.need(4); c.h_bad.need(4);
    {   // (a param that is used incrementally: fronts near the root collect the rows of every loop closure since the plan was made and
        // may outgrow the single-workgroup kernel -- room for a few of them on the multi-workgroup path, whose scratch a plan without
        // such fronts would not have; measured on the M3500 demo: 13 steps re-planned for 70 KB of scratch)
        size_t mx = inc ? diag_doubles(16, 512) : 1;
        for (int l = 0; l < P.nLevels; l++) mx = std::max(mx, diag_doubles(c.levels[l].n_big, c.levels[l].n_diag_slots));
        c.d_diag.need(mx);
    }
    c.st.n_fronts = P.nF; c.st.n_levels = P.nLevels; c.st.max_front_rows = P.max_rows;
    c.st.nnz_L = P.nnzL; c.st.flops_factor = P.flops; c.st.bytes_fronts = 8.0 * (double)pool_doubles;
}

static void set_small_attr() {
    static std::once_flag once;
    std::call_once(once, [] {
        HIPCHECK(hipFuncSetAttribute((const void *)k_front_small<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_front_small<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_front_small<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_backsolve_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_backsolve_t<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_backsolve_w, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_backsolve_t<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_block_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_backsolve_blk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_assemble_tile, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_block_solve<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_block_solve<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_inc_one<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_inc_one<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_inc_one<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    });
}

// back substitution of one level: update-row products of the large fronts on many workgroups, then one workgroup per front
template <class Tic, class Toc>
static void launch_backsolve(Context &c, const LevelPlan &L, hipStream_t s, Tic tic, Toc toc, const int *tab = nullptr, UpdArgs upd = UpdArgs{}) {
    if (!tab) tab = c.d_tab.p;
    if (!L.n_all) return;
    tic(K_BACKSOLVE);
    if (L.bs_gemv.grid > 0)
        hipLaunchKernelGGL(k_backsolve_gemv, dim3(L.bs_gemv.grid), dim3(TPB), 0, s, c.dp, tab + L.bs_gemv.list_off, tab + L.bs_gemv.pre_off,
                           L.bs_gemv.n, c.d_pool.p, c.d_x.p);
    // wide fronts: chain + helper workgroups (k_backsolve_blk); the level's other fronts below
    int n_all = L.n_all, all_off = L.all_off; size_t solve_lds = L.solve_lds;
    if (L.bs_blk.grid > 0) {
        hipLaunchKernelGGL(k_backsolve_blk, dim3(L.bs_blk.grid), dim3(TPB), L.bs_blk_lds, s, c.dp, tab + L.bs_blk.list_off, tab + L.bs_blk.pre_off, L.bs_blk.n,
                           c.d_pool.p, c.d_x.p, c.d_dinv.p, c.d_bsb_flags.p, c.d_bsb_far.p, L.bs_gemv.grid > 0 ? 1 : 0, c.d_bad.p, upd);
        n_all = L.n_rest; all_off = L.rest_off; solve_lds = L.rest_lds;
        if (!n_all) { toc(); return; }
    }
    // latency-bound levels of small fronts: column-per-lane form with the L panel in LDS (at least two workgroups per CU)
    if (g_opt.wave_backsolve && L.bs_gemv.grid == 0 && L.n_all < g_opt.tp_fronts && L.maxns <= BSW_MAX_NS && L.solve_w_lds <= 80 * 1024)
        hipLaunchKernelGGL(k_backsolve_w, dim3(n_all), dim3(TPB), L.solve_w_lds, s, c.dp, tab + all_off, c.d_pool.p, c.d_x.p, (int *)nullptr, c.d_bad.p, upd);
    else if (solve_lds >= (size_t)(BS_TALL_ROWS + NB + 8 + NB * (NB + 1)) * 8)
        hipLaunchKernelGGL((k_backsolve_t<false, true>), dim3(n_all), dim3(TPB), solve_lds, s, c.dp, tab + all_off, c.d_pool.p, c.d_x.p, L.bs_gemv.grid > 0 ? 1 : 0, (int *)nullptr, 0, c.d_bad.p, upd);
    else
        hipLaunchKernelGGL((k_backsolve_t<false>), dim3(n_all), dim3(TPB), solve_lds, s, c.dp, tab + all_off, c.d_pool.p, c.d_x.p, L.bs_gemv.grid > 0 ? 1 : 0, (int *)nullptr, 0, c.d_bad.p, upd);
    toc();
}

// k_front_small with the configured workgroup size (option small_threads: 256 / 512 / 1024)
// the multi-level launch of the factorisation: every small front of levels >= persist_l0
static void launch_front_persist(Context &c, hipStream_t s) {
    const int *list = c.d_tab.p + c.p_up_off;
    int *fl = c.d_flags.p;
    if (c.p_nt >= 1024) hipLaunchKernelGGL(k_front_small<1024>, dim3(c.p_up_n), dim3(1024), c.p_up_lds, s, c.dp, list, c.d_pool.p, c.d_H.p, c.d_bad.p, c.p_up_full, g_opt.block_factor, fl, 1);
    else if (c.p_nt >= 512) hipLaunchKernelGGL(k_front_small<512>, dim3(c.p_up_n), dim3(512), c.p_up_lds, s, c.dp, list, c.d_pool.p, c.d_H.p, c.d_bad.p, c.p_up_full, g_opt.block_factor, fl, 1);
    else hipLaunchKernelGGL(k_front_small<256>, dim3(c.p_up_n), dim3(256), c.p_up_lds, s, c.dp, list, c.d_pool.p, c.d_H.p, c.d_bad.p, c.p_up_full, g_opt.block_factor, fl, 1);
}

static void launch_front_small(Context &c, const LevelPlan &L, hipStream_t s, const int *tab = nullptr) {
    if (!tab) tab = c.d_tab.p;
    const int nt = L.small_nt;
    if (nt >= 1024) hipLaunchKernelGGL(k_front_small<1024>, dim3(L.n_small), dim3(1024), L.small_lds, s, c.dp, tab + L.small_off, c.d_pool.p, c.d_H.p, c.d_bad.p, L.full_limit, g_opt.block_factor, (int *)nullptr, 0);
    else if (nt >= 512) hipLaunchKernelGGL(k_front_small<512>, dim3(L.n_small), dim3(512), L.small_lds, s, c.dp, tab + L.small_off, c.d_pool.p, c.d_H.p, c.d_bad.p, L.full_limit, g_opt.block_factor, (int *)nullptr, 0);
    else hipLaunchKernelGGL(k_front_small<256>, dim3(L.n_small), dim3(256), L.small_lds, s, c.dp, tab + L.small_off, c.d_pool.p, c.d_H.p, c.d_bad.p, L.full_limit, g_opt.block_factor, (int *)nullptr, 0);
}

// panel steps of the big fronts of one level: per NB-column panel {diagonal block, row solves, narrow update};
// after every OBP panels one wide update with K = OBP * NB (kernels.hip.h: syrk_range)
template <class Tic, class Toc>
static void enqueue_big_steps(Context &c, const LevelPlan &L, hipStream_t s, Tic tic, Toc toc, bool la = false, const int *tab = nullptr) {
    if (!tab) tab = c.d_tab.p;
    auto wide = [&](const Launch &w, int k, int mode, hipStream_t st) {
        if (w.tile == TILE2)
            hipLaunchKernelGGL(k_syrk_big128, dim3(w.grid), dim3(TPB), 0, st, c.dp, tab + w.list_off, tab + w.pre_off, w.n, k / OBP * OBP, k + 1, mode, c.d_pool.p);
        else
            hipLaunchKernelGGL(k_syrk_big, dim3(w.grid), dim3(TPB), 0, st, c.dp, tab + w.list_off, tab + w.pre_off, w.n, k / OBP * OBP, k + 1, mode, c.d_pool.p);
    };
    if (la && !c.s2) {        // lowest priority: its big kernels must not delay the one-workgroup kernels of the chain
        int lo = 0, hi = 0;
        HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHECK(hipStreamCreateWithPriority(&c.s2, hipStreamNonBlocking, lo));
    }
    if (g_opt.block_panels) {
        // outer-block panels: per 128-column outer block {diagonal block in LDS, row solves on the matrix cores, wide update}.
        // Look-ahead (la): the wide update is split -- "ahead" = the next outer block's columns, on this stream, all the next
        // diagonal block and row solves need; "rest" = everything right of them, on the side stream beside that chain.  Both
        // write disjoint columns; the next "ahead" and "rest" touch columns the previous "rest" wrote, so they wait for it.
        const int steps = (int)L.panel.size();
        hipEvent_t rest_done = nullptr;
        for (size_t o = 0; o < L.bchain.size(); o++) {
            const Launch &bc = L.bchain[o], &bt = L.btile[o];
            tic(K_PANEL_BIG);
            hipLaunchKernelGGL(k_block_chain, dim3(bc.n), dim3(BCH_THREADS), block_chain_lds(), s, c.dp, tab + bc.list_off, (int)o, c.d_pool.p, c.d_diag.p, c.d_dinv.p, c.d_bad.p);
            if (bt.grid > 0) {
                if (bt.tile == 2) hipLaunchKernelGGL(k_block_solve<2>, dim3(bt.grid), dim3(TPB), block_solve_lds(), s, c.dp, tab + bt.list_off, tab + bt.pre_off, bt.n, (int)o, c.d_pool.p, c.d_diag.p, c.d_dinv.p);
                else hipLaunchKernelGGL(k_block_solve<1>, dim3(bt.grid), dim3(TPB), block_solve_lds(), s, c.dp, tab + bt.list_off, tab + bt.pre_off, bt.n, (int)o, c.d_pool.p, c.d_diag.p, c.d_dinv.p);
            }
            toc();
            const int k = std::min((int)(o + 1) * OBP, steps) - 1;      // the panel step that closes the outer block carries its wide update
            if (!la) {
                const Launch &sw = L.syrkw[k];
                if (sw.grid > 0) { tic(K_SYRK_BIG); wide(sw, k, 1, s); toc(); }
                continue;
            }
            const Launch &sa = L.syrka[k], &sb = L.syrkb[k];
            if (sb.grid > 0) {
                hipEvent_t solved = c.la_event();
                HIPCHECK(hipEventRecord(solved, s));
                HIPCHECK(hipStreamWaitEvent(c.s2, solved, 0));
                wide(sb, k, 3, c.s2);                       // (same stream as the previous "rest": in order behind it)
            }
            if (sa.grid > 0) {
                if (rest_done) HIPCHECK(hipStreamWaitEvent(s, rest_done, 0));
                wide(sa, k, 2, s);
            }
            if (sb.grid > 0) { rest_done = c.la_event(); HIPCHECK(hipEventRecord(rest_done, c.s2)); }
        }
        if (rest_done) HIPCHECK(hipStreamWaitEvent(s, rest_done, 0));      // join: the next level reads the update blocks
        return;
    }
    hipEvent_t rest_done = nullptr;          // completion of the latest "rest" update on the side stream
    for (size_t k = 0; k < L.panel.size(); k++) {
        const Launch &pa = L.panel[k], &sy = L.syrk[k], &sw = L.syrkw[k];
        const bool ll = g_opt.left_panels && g_opt.fused_panel;       // left-looking panels: no narrow update launches
        if (ll) {
            tic(K_PANEL_BIG);
            hipLaunchKernelGGL(k_diagpanel_ll, dim3(pa.grid), dim3(TPB), 0, s, c.dp, tab + pa.list_off, tab + pa.pre_off, pa.n, (int)k, c.d_pool.p,
                               c.d_diag.p, L.diag_slot0[k], c.d_bad.p);
            toc();
        } else if (pa.single) {          // one row tile per front: diagonal block + row solves in one launch
            tic(K_PANEL_BIG);
            hipLaunchKernelGGL(k_diagpanel_big, dim3(pa.n), dim3(TPB), 0, s, c.dp, tab + pa.list_off, (int)k, c.d_pool.p, c.d_bad.p);
            toc();
        } else if (g_opt.fused_panel) {   // several row tiles per front: every tile factors the diagonal block itself, one launch
            tic(K_PANEL_BIG);
            hipLaunchKernelGGL(k_diagpanel_multi, dim3(pa.grid), dim3(TPB), 0, s, c.dp, tab + pa.list_off, tab + pa.pre_off, pa.n, (int)k, c.d_pool.p,
                               c.d_diag.p, L.diag_slot0[k], c.d_bad.p);
            toc();
        } else {
            tic(K_DIAG_BIG);
            hipLaunchKernelGGL(k_diag_big, dim3(pa.n), dim3(64), 0, s, c.dp, tab + pa.list_off, (int)k, c.d_pool.p, c.d_diag.p, c.d_bad.p);
            toc();
            tic(K_PANEL_BIG);
            hipLaunchKernelGGL(k_panel_big, dim3(pa.grid), dim3(TPB), 0, s, c.dp, tab + pa.list_off, tab + pa.pre_off, pa.n, (int)k, c.d_pool.p, c.d_diag.p);
            toc();
        }
        if (sy.grid > 0 && !ll) {
            tic(K_SYRK_BIG);
            hipLaunchKernelGGL(k_syrk_big, dim3(sy.grid), dim3(TPB), 0, s, c.dp, tab + sy.list_off, tab + sy.pre_off, sy.n, (int)k, (int)k + 1, 0, c.d_pool.p);
            toc();
        }
        if (!la) {
            if (sw.grid > 0) { tic(K_SYRK_BIG); wide(sw, (int)k, 1, s); toc(); }
            continue;
        }
        // Look-ahead.  "ahead" = the next outer block's panel columns: stays on this stream, the chain of small kernels
        // that follows needs it.  "rest" = everything right of them: side stream, overlapped with that chain.  Both read
        // this outer block's columns and write disjoint column ranges.  The previous "rest" wrote the columns "ahead"
        // updates now (and the ones this "rest" updates: same stream, in order), so "ahead" waits for it.
        const Launch &sa = L.syrka[k], &sb = L.syrkb[k];
        if (sb.grid > 0) {
            hipEvent_t chain_done = c.la_event();
            HIPCHECK(hipEventRecord(chain_done, s));
            HIPCHECK(hipStreamWaitEvent(c.s2, chain_done, 0));
            wide(sb, (int)k, 3, c.s2);
        }
        if (sa.grid > 0) {
            if (rest_done) HIPCHECK(hipStreamWaitEvent(s, rest_done, 0));
            wide(sa, (int)k, 2, s);
        }
        if (sb.grid > 0) { rest_done = c.la_event(); HIPCHECK(hipEventRecord(rest_done, c.s2)); }
    }
    if (rest_done) HIPCHECK(hipStreamWaitEvent(s, rest_done, 0));      // join: the next level reads the update blocks
    if (g_opt.fused_panel && L.n_wb > 0) {       // the diagonal blocks parked by k_diagpanel_multi go into their fronts
        tic(K_DIAG_BIG);
        hipLaunchKernelGGL(k_diag_writeback, dim3(L.n_wb), dim3(TPB), 0, s, c.dp, tab + L.wb_off, c.d_pool.p, c.d_diag.p);
        toc();
    }
}

// kernels of one level of the factorisation (small LDS fronts, big multi-workgroup path)
template <class Tic, class Toc>
static void enqueue_factor_level(Context &c, const LevelPlan &L, hipStream_t s, Tic tic, Toc toc, bool la = false, const int *tab = nullptr) {
    if (!tab) tab = c.d_tab.p;
    if (L.n_small) {
        tic(K_FRONT_SMALL);
        launch_front_small(c, L, s, tab);
        toc();
    }
    if (L.n_big) {
        tic(K_ASSEMBLE_BIG);
        if (g_opt.tile_assembly)
            hipLaunchKernelGGL(k_assemble_tile, dim3(L.asm_tile.grid), dim3(TPB), at_lds(), s, c.dp, tab + L.asm_tile.list_off,
                               tab + L.asm_tile.pre_off, L.asm_tile.n, c.d_pool.p, c.d_H.p);
        else
            hipLaunchKernelGGL(k_assemble_big, dim3(L.asm_big.grid), dim3(TPB), L.asm_lds, s, c.dp, tab + L.asm_big.list_off,
                               tab + L.asm_big.pre_off, L.asm_big.n, c.d_pool.p, c.d_H.p);
        toc();
        enqueue_big_steps(c, L, s, tic, toc, la, tab);
    }
}

// enqueue: linearise -> per level {assemble+factor} -> back substitution -> state update
// ev != null: record stage events (0 start, 1 after linearise, 2 after factor, 3 after solve+update)
// ktime: bracket EVERY kernel launch with its own HIP event pair on this stream (c.k_ev / c.k_ids)
static void enqueue_numeric(Context &c, GraphPack &gp, hipStream_t s, hipEvent_t *ev, bool unary_at_lp = false, bool ktime = false, bool io_host = false) {
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
    if (c.dp.prof) HIPCHECK(hipMemsetAsync(c.d_prof.p, 0, (size_t)8 * PROF_SLOTS * P.nF, s));
    if (io_host) hipLaunchKernelGGL(k_load_states, dim3((3 * N + TPB - 1) / TPB), dim3(TPB), 0, s, 3 * N, gp.h_state.p, gp.d_state.p, gp.d_lp.p);
    tic(K_LINEARIZE);
    if (F >= g_opt.linearize_staged_min)
        hipLaunchKernelGGL((k_linearize_t<true>), dim3((F + TPB - 1) / TPB), dim3(TPB), 0, s, 0, F, (const int *)nullptr, gp.d_fa.p, gp.d_fb.p, gp.d_z.p, gp.d_W.p,
                           gp.d_lp.p, gp.d_state.p, c.d_swap.p, c.dp.slot_blk, c.dp.slot_rhs, c.d_H.p, c.d_bad.p, unary_at_lp ? gp.d_upt.p : (const double *)nullptr,
                           P.nF, c.d_flevel.p, c.persist_l0 >= 0 ? c.persist_l0 : 0, c.persist_l0 >= 0 ? c.d_flags.p : (int *)nullptr);
    else
        hipLaunchKernelGGL((k_linearize_t<false>), dim3((F + TPB - 1) / TPB), dim3(TPB), 0, s, 0, F, (const int *)nullptr, gp.d_fa.p, gp.d_fb.p, gp.d_z.p, gp.d_W.p,
                           gp.d_lp.p, gp.d_state.p, c.d_swap.p, c.dp.slot_blk, c.dp.slot_rhs, c.d_H.p, c.d_bad.p, unary_at_lp ? gp.d_upt.p : (const double *)nullptr,
                           P.nF, c.d_flevel.p, c.persist_l0 >= 0 ? c.persist_l0 : 0, c.persist_l0 >= 0 ? c.d_flags.p : (int *)nullptr);
    if (!gp.host_idx.empty()) {         // host-evaluated factors: their blocks replace the null contributions written above
        const int nh = (int)gp.host_idx.size();
        HIPCHECK(hipMemcpyAsync(gp.d_hostH.p, gp.h_hostH.p, (size_t)33 * 8 * nh, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_scatter_host, dim3((nh + TPB - 1) / TPB), dim3(TPB), 0, s, nh, gp.d_host_idx.p, gp.d_hostH.p, gp.d_fb.p, c.d_swap.p,
                           c.dp.slot_blk, c.dp.slot_rhs, c.d_H.p);
    }
    toc();
    if (ev) HIPCHECK(hipEventRecord(ev[1], s));
    c.la_next = 0;
    const int l0 = c.persist_l0 >= 0 ? c.persist_l0 : P.nLevels;        // levels >= l0: one multi-level launch each way
    for (int l = 0; l < l0; l++) enqueue_factor_level(c, c.levels[l], s, tic, toc, !ktime && g_opt.lookahead);
    if (l0 < P.nLevels) { tic(K_FRONT_SMALL); launch_front_persist(c, s); toc(); }
    if (ev) HIPCHECK(hipEventRecord(ev[2], s));
    // the state update of a front's own poses rides on its back substitution (no kernel of its own); the last launch also
    // mirrors the pivot flag for the API call
    UpdArgs upd{ c.d_perm.p, gp.d_lp.p, gp.d_state.p, gp.d_dx.p, io_host ? gp.h_lp.p : nullptr, io_host ? gp.h_dx.p : nullptr, nullptr };
    if (l0 < P.nLevels) {
        UpdArgs u = upd; if (l0 == 0) u.bad_out = io_host ? c.h_bad.p : nullptr;
        tic(K_BACKSOLVE);
        if (g_opt.wave_backsolve && c.p_dn_maxns <= BSW_MAX_NS)
            hipLaunchKernelGGL(k_backsolve_w, dim3(c.p_dn_n), dim3(TPB), c.p_dn_lds, s, c.dp, c.d_tab.p + c.p_dn_off, c.d_pool.p, c.d_x.p, c.d_flags.p + P.nF, c.d_bad.p, u);
        else
            hipLaunchKernelGGL((k_backsolve_t<true>), dim3(c.p_dn_n), dim3(TPB), c.p_dn_lds, s, c.dp, c.d_tab.p + c.p_dn_off, c.d_pool.p, c.d_x.p, 0, c.d_flags.p + P.nF, 1, c.d_bad.p, u);
        toc();
    }
    for (int l = l0 - 1; l >= 0; l--) {
        UpdArgs u = upd; if (l == 0) u.bad_out = io_host ? c.h_bad.p : nullptr;
        launch_backsolve(c, c.levels[l], s, tic, toc, nullptr, u);
    }
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
static void run_numeric(Context &c, GraphPack &gp, bool timing, bool unary_at_lp = false, bool io_host = false) {
    hipStream_t s = gp.stream;
    set_small_attr();
    if (timing && !c.have_events) { for (auto &e : c.ev) HIPCHECK(hipEventCreate(&e)); c.have_events = true; }
    if (io_host) {
        if (g_opt.use_graph && !timing && gp.host_idx.empty()) {
            const void *key[7] = { gp.d_state.p, gp.h_state.p, gp.h_lp.p, gp.h_dx.p, c.h_bad.p, (const void *)(size_t)gp.N, (const void *)(size_t)gp.serial };
            if (!c.gexec_api || memcmp(key, c.api_key, sizeof(key)) != 0) {
                if (c.gexec_api) { (void)hipGraphExecDestroy(c.gexec_api); c.gexec_api = nullptr; }
                hipGraph_t graph = nullptr;
                HIPCHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                enqueue_numeric(c, gp, s, nullptr, false, false, true);
                HIPCHECK(hipStreamEndCapture(s, &graph));
                HIPCHECK(hipGraphInstantiate(&c.gexec_api, graph, nullptr, nullptr, 0));
                HIPCHECK(hipGraphDestroy(graph));
                memcpy(c.api_key, key, sizeof(key));
            }
            HIPCHECK(hipGraphLaunch(c.gexec_api, s));
        } else {
            enqueue_numeric(c, gp, s, timing ? c.ev : nullptr, false, false, true);
        }
        return;
    }
    if (g_opt.use_graph && !timing && !unary_at_lp && gp.host_idx.empty()) {   // (host-evaluated factors: staging buffers may move)
        if (!c.gexec || c.gexec_key != (const void *)gp.d_state.p || c.gexec_serial != gp.serial) {
            if (c.gexec) { (void)hipGraphExecDestroy(c.gexec); c.gexec = nullptr; }
            hipGraph_t graph = nullptr;
            HIPCHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            enqueue_numeric(c, gp, s, nullptr);
            HIPCHECK(hipStreamEndCapture(s, &graph));
            HIPCHECK(hipGraphInstantiate(&c.gexec, graph, nullptr, nullptr, 0));
            HIPCHECK(hipGraphDestroy(graph));
            c.gexec_key = (const void *)gp.d_state.p; c.gexec_serial = gp.serial;
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

// Options are baked into the launch tables at plan time (build_level: small / panel / big classification, tile counts,
// diagonal-block slots) AND read again when the kernels are enqueued or captured into a hipGraph.  Every change of an option
// that touches either (api_set_option bumps g_opt_epoch) therefore forces a re-plan and a re-capture on every param.
static long long g_opt_epoch = 0;
static long long launch_table_key() { return g_opt_epoch; }
// make sure plan / device buffers match the packed graph; returns true if the plan was reused
static bool prepare_plan(Context &c, GraphPack &gp, const april_graph_t *g, bool upload = true) {
    const int N = gp.N, F = gp.F;
    bool same = c.have_plan && c.patN == N && (int)c.pat.size() == 2 * F && c.plan.leaf_nodes == g_opt.leaf_nodes && c.plan_pin == g_opt.pin_last &&
                c.plan_persist == launch_table_key() && c.inc.t_first.empty();        // (a plan extended by tail fronts is only driven by inc_fast_step)
    if (same) {
        for (int i = 0; i < F && same; i++) same = c.pat[2 * i] == gp.h_fa.p[i] && c.pat[2 * i + 1] == gp.h_fb.p[i];
    }
    if (same) return true;
    c.pat.resize((size_t)2 * F);
    for (int i = 0; i < F; i++) { c.pat[2 * i] = gp.h_fa.p[i]; c.pat[2 * i + 1] = gp.h_fb.p[i]; }
    c.patN = N; c.plan_pin = g_opt.pin_last; c.plan_persist = launch_table_key();
    std::vector<double> xy((size_t)2 * N);
    for (int i = 0; i < N; i++) { xy[2 * i] = gp.h_state.p[3 * i]; xy[2 * i + 1] = gp.h_state.p[3 * i + 1]; }
    const double tb0 = now_ms();
    build_plan(c.plan, N, F, c.pat.data(), xy.data(), g_opt.leaf_nodes);
    const double tb1 = now_ms();
    if (upload) upload_plan(c, gp.stream);
    if (getenv("APRILSAM_AMD_PLAN_PROFILE")) fprintf(stderr, "aprilsam_amd plan: N=%d build %.3f ms upload %.3f ms\n", N, tb1 - tb0, now_ms() - tb1);
    c.have_plan = true;
    return false;
}

// ------------------------------------------------------------------------------------------------------
// incremental fast path: frozen base plan + a chain of small TAIL fronts + regeneration of the dirty root paths only.
//
// The plan of the last batch step stays frozen.  Poses added since are eliminated after every base pose, in id order,
// grouped into tail fronts of at most TAIL_POSES poses (front ids nF0, nF0 + 1, ...; only the last one grows).  Every
// front carries E = the tail poses in its structure beyond its own columns (base fronts: appended behind their frozen
// base structure).  A new factor makes its owner front dirty and pushes its later endpoint into E along the assembly
// path up to the front that owns it; a front whose E changed, that owns a new factor or that has a dirty child is
// regenerated (descriptor, destination records with indirect source lists, child maps -- appended to device arenas
// reserved at plan upload) and re-factorised; clean fronts keep their factors and Schur updates in HBM.  Small tail
// fronts keep every regenerated front inside the single-workgroup LDS kernel: one launch per dirty front on the root
// path instead of the multi-launch big-front path one ever-growing tail front ran into.
// ------------------------------------------------------------------------------------------------------

static void inc_prepare(Context &c) {        // after a full (re)plan: c.plan is the new base
    IncState &I = c.inc; const Plan &P = c.plan;
    I.Nb = P.N; I.Fb = P.F; I.nF0 = P.nF; I.nLev0 = P.nLevels;
    I.pos_front.assign(P.N, 0);
    for (int t = 0; t < P.nF; t++) for (int k = 0; k < P.f_nsb[t]; k++) I.pos_front[P.f_first[t] + k] = t;
    I.parent.assign(P.f_parent.begin(), P.f_parent.end());
    I.E.assign(P.nF, {}); I.xfac.assign(P.nF, {});
    I.bf_ptr.assign(P.nF + 1, 0);
    for (int f = 0; f < P.F; f++) if (P.fac_front[f] >= 0) I.bf_ptr[P.fac_front[f] + 1]++;
    for (int t = 0; t < P.nF; t++) I.bf_ptr[t + 1] += I.bf_ptr[t];
    I.bf_idx.resize(P.F);
    { std::vector<int> fill(I.bf_ptr.begin(), I.bf_ptr.end() - 1); for (int f = 0; f < P.F; f++) if (P.fac_front[f] >= 0) I.bf_idx[fill[P.fac_front[f]]++] = f; }
    I.rel_begin.assign(P.nF, 0); I.cur_nub.assign(P.nF, 0); I.cur_cap.assign(P.nF, 0);
    for (int t = 0; t < P.nF; t++) { I.rel_begin[t] = (int)(I.o_rel + P.f_rows_ptr[t]); I.cur_nub[t] = P.f_nub[t]; I.cur_cap[t] = (long long)P.rows(t) * P.cols(t); }
    I.dirty.assign(P.nF, 0);
    I.f_level.assign(P.f_level.begin(), P.f_level.end());
    I.fd.resize(P.nF);
    I.t_first.clear(); I.t_cnt.clear(); I.tf_of.clear(); I.kids.assign(P.nF, {}); I.tail_ok = -1; I.recs_stale = -1;
    I.stale.assign(P.nF, 0); I.upd_ok = true;
    I.base_has_big = false;
    for (const LevelPlan &L : c.levels) I.base_has_big = I.base_has_big || L.n_big > 0;
    I.base_levels = c.levels;
    c.inc_slot_blk.clear(); c.inc_slot_rhs.clear();
    I.ready = true;
}

// Regenerate the dirty part of the plan for nodes [Nold, N) / factors [Fold, F) and run the numeric phase on it.
// Returns false (nothing enqueued) when the step does not fit the frozen structure or the reserved slack.
// batch_lambda >= 0: the same structures driven as a BATCH step (april_graph_cholesky on a graph that only grew since the
// plan was made): every node re-linearised, every factor linearised, the Tikhonov term batch_lambda on every pose,
// every front -- base and tail -- re-factorised, full back substitution.  Saves the nested dissection + symbolic analysis
// + plan upload (6-7 ms on M3500) that a cold call pays, at the price of a less bushy tree for the appended poses.
static bool inc_fast_step(Context &c, GraphPack &gp, int N, int F, int Fold, int Nold, const std::vector<RefModel::Visit> *needed, double batch_lambda = -1.0,
                          bool patch_states = false) {
    IncState &I = c.inc; Plan &P = c.plan;
    const bool batch = batch_lambda >= 0;
    if (!I.ready || N < I.Nb || Fold < I.Fb) return inc_fail(1);
    // an option that launch tables, front layouts (tail_poses: the padded shape of the last tail front) or captured graphs depend
    // on changed since this plan was made: the frozen base + tail structures were built under the old values -- full re-plan
    if (c.plan_persist != launch_table_key()) return inc_fail(2);
    const int Nb = I.Nb, nF0 = I.nF0, m = N - Nb;
    if (m > I.cap_nodes - 8 || F - I.Fb > I.cap_fact - 8 || m < 1) return inc_fail(3);
    const int *fa = gp.h_fa.p, *fb = gp.h_fb.p;
    hipStream_t s = gp.stream;
    auto local_base = [&](int t, int p) -> int {      // local block index of base position p in base front t, or -1
        if (p >= P.f_first[t] && p < P.f_first[t] + P.f_nsb[t]) return p - P.f_first[t];
        const int *b = P.f_rows.data() + P.f_rows_ptr[t], *e = b + P.f_nub[t];
        const int *it = std::lower_bound(b, e, p);
        return (it == e || *it != p) ? -1 : P.f_nsb[t] + (int)(it - b);
    };
    // The commonest step -- new poses and factors among the last few poses of the LAST tail front -- re-factorises that front's
    // trailing columns only (tail_refactor, kernels.hip.h): the front's factor on the device must be complete, its array
    // keeps its shape (phantom rows, below), and nothing else may be touched by the step's factors.
    bool tail_fast = false; TailStep tstep{ -1, 0, 0, 0, 0, 0, 0 };
    if (!batch && g_opt.inc_tail && g_opt.inc_multi && g_opt.persist && !I.t_first.empty() && I.tail_ok == nF0 + (int)I.t_first.size() - 1 && F > Fold) {
        const int first = I.t_first.back(), n_old = I.t_cnt.back(), n_new = n_old + (N - std::max(Nold, Nb));
        int lo = first + n_old;                        // (poses added by this step: all of their columns are new)
        bool ok = n_new <= TAIL_POSES && I.E.back().empty() && F - Fold <= TAIL_MAXF;
        for (int f = Fold; f < F && ok; f++) {
            const int a = fa[f], b = fb[f];
            ok = a >= first && (b < 0 || b >= first);
            lo = std::min(lo, b >= 0 ? std::min(a, b) : a);
        }
        if (ok && n_new - (lo - first) <= TAILK) {
            const FrontDesc &Dt = I.fd[nF0 + (int)I.t_first.size() - 1];      // (its shape stays: nsb + nub = the padded capacity)
            tail_fast = true; tstep = TailStep{ nF0 + (int)I.t_first.size() - 1, lo - first, n_old, n_new, Dt.nsb + Dt.nub, first, Dt.off };
        }
    }
    // ... and when the reference's walk stays on a short root path, the whole step is decided here, without the general
    // machinery below (whose cost grows with the number of fronts and levels): one k_inc_one launch
    g_incsub[6] += tail_fast ? 1 : 0; g_incsub[7] += (tail_fast && needed && patch_states) ? 1 : 0;
    if (tail_fast && needed && g_opt.inc_one && g_opt.wave_backsolve && patch_states && F <= gp.F_cap) {
        const int T = tstep.t, first = I.t_first.back(), nT0 = (int)I.t_first.size(), nFr0 = nF0 + nT0;
        const int n_new = I.t_cnt.back() + (N - std::max(Nold, Nb)), nph = TAIL_POSES - n_new;
        auto nsb_now = [&](int t) { return t == T ? n_new : (t >= nF0 ? I.t_cnt[t - nF0] : P.f_nsb[t]); };
        auto nub_now = [&](int t) { return t == T ? nph : I.cur_nub[t]; };
        // the fronts of the visited poses and their ancestors, top level first
        std::vector<int> &lst = I.st_ids; lst.clear();
        I.need.assign(nFr0, 0);
        bool fits = true; size_t lds = tail_refactor_lds(); int maxns = 0;
        for (const RefModel::Visit &v : *needed) {
            int t = v.node >= first ? T : (v.node >= Nb ? I.tf_of[v.node - Nb] : I.pos_front[P.pos[v.node]]);
            while (t >= 0 && !I.need[t]) { I.need[t] = 1; lst.push_back(t); t = I.parent[t]; }
        }
        // every visited pose inside the window of trailing columns the kernel holds in LDS anyway (widened to the first visited
        // pose): the back substitution and the state update happen right there (tail_refactor, ts.solve) -- no pass over the
        // whole front, no front list
        bool solve_here = false;
        if (g_opt.inc_tail_solve && !needed->empty() && g_opt.inc_one_threads < 1024) {
            int v_lo = 1 << 30;
            for (const RefModel::Visit &v : *needed) v_lo = std::min(v_lo, v.node);
            const int lo_all = std::min(first + tstep.a_idx, v_lo);
            if (v_lo >= first && n_new - (lo_all - first) <= TAILK) { solve_here = true; tstep.s_idx = lo_all - first; tstep.solve = 1; lst.clear(); lst.push_back(T); }
        }
        fits = !lst.empty() && (int)lst.size() <= g_opt.inc_one_dn;
        if (fits && solve_here) fits = (size_t)9 * (I.slots_used + 5 * (F - Fold)) <= c.d_H.cap && (size_t)N <= c.d_perm.cap;
        else if (fits) {
            std::sort(lst.begin(), lst.end(), [&](int x, int y) { return I.f_level[x] != I.f_level[y] ? I.f_level[x] > I.f_level[y] : x < y; });
            for (int t : lst) { lds = std::max(lds, backsolve_lds(3 * (nsb_now(t) + nub_now(t)), 3 * nsb_now(t), true)); maxns = std::max(maxns, 3 * nsb_now(t)); }
            fits = lds <= 160 * 1024 && maxns <= BSW_MAX_NS && I.tab_used + (long long)lst.size() <= (long long)c.d_tab.cap &&
                   (size_t)9 * (I.slots_used + 5 * (F - Fold)) <= c.d_H.cap && (size_t)N <= c.d_perm.cap;
        }
        if (fits) {
            // bookkeeping, as sections 0-2 below do it for this case
            for (int k = std::max(Nold, Nb); k < N; k++) { I.t_cnt.back()++; I.tf_of.push_back(T); }
            std::vector<int> &new_slot_blk = I.st_sb, &new_slot_rhs = I.st_sr; std::vector<unsigned char> &new_swap = I.st_sw;
            new_slot_blk.resize((size_t)3 * (F - Fold)); new_slot_rhs.resize((size_t)2 * (F - Fold)); new_swap.resize(F - Fold);
            c.inc_slot_blk.resize((size_t)3 * (F - I.Fb), -1); c.inc_slot_rhs.resize((size_t)2 * (F - I.Fb), -1);
            for (int f = Fold; f < F; f++) {
                I.xfac[T].push_back(f);
                for (int k = 0; k < 3; k++) new_slot_blk[(size_t)3 * (f - Fold) + k] = c.inc_slot_blk[(size_t)3 * (f - I.Fb) + k] = I.slots_used++;
                for (int k = 0; k < 2; k++) new_slot_rhs[(size_t)2 * (f - Fold) + k] = c.inc_slot_rhs[(size_t)2 * (f - I.Fb) + k] = I.slots_used++;
                new_swap[f - Fold] = fb[f] >= 0 && fa[f] < fb[f];
            }
            FrontDesc &D = I.fd[T];
            D.nsb = n_new; D.nub = nph; I.cur_nub[T] = nph;
            I.recs_stale = T; I.tail_ok = T;
            c.st.reserved0 = 1; c.st.inc_fronts_updated = 0;
            // patches
            PatchList &PL = c.patches;
            PL.reset();
            const int f0 = gp.F_on_device;
            if (F > f0) {
                PL.add(gp.d_fa.p + f0, gp.h_fa.p + f0, (size_t)(F - f0) * 4); PL.add(gp.d_fb.p + f0, gp.h_fb.p + f0, (size_t)(F - f0) * 4);
                PL.add(gp.d_z.p + (size_t)3 * f0, gp.h_z.p + (size_t)3 * f0, (size_t)(F - f0) * 24);
                PL.add(gp.d_W.p + (size_t)9 * f0, gp.h_W.p + (size_t)9 * f0, (size_t)(F - f0) * 72);
                gp.F_on_device = F;
            }
            if (!solve_here) PL.add(c.d_tab.p + I.tab_used, lst.data(), lst.size() * 4);
            PL.add(c.d_fd.p + T, &D, sizeof(FrontDesc));
            PL.add((int *)c.dp.slot_blk + (size_t)3 * Fold, new_slot_blk.data(), new_slot_blk.size() * 4);
            PL.add((int *)c.dp.slot_rhs + (size_t)2 * Fold, new_slot_rhs.data(), new_slot_rhs.size() * 4);
            PL.add(c.d_swap.p + Fold, new_swap.data(), new_swap.size());
            if (N > Nold) {
                int ids[TAILK + 1]; double zeros[TAILK + 1];
                const int nn = N - Nold;              // (<= TAILK: the new poses are among the trailing columns)
                for (int i = 0; i < nn; i++) { ids[i] = Nold + i; zeros[i] = 0.0; }
                PL.add(c.d_pos.p + Nold, ids, (size_t)nn * 4);
                PL.add(c.d_perm.p + Nold, ids, (size_t)nn * 4);
                PL.add(c.d_lambda.p + Nold, zeros, (size_t)nn * 8);       // no Tikhonov term on poses added incrementally (aprilsam.c:508-542)
                c.lambda_N = -1;
                P.perm.resize(N); P.pos.resize(N);
                for (int i = Nold; i < N; i++) { P.perm[i] = i; P.pos[i] = i; }
            }
            for (int i : gp.changed) {
                PL.add(gp.d_state.p + (size_t)3 * i, gp.h_state.p + (size_t)3 * i, 24);
                PL.add(gp.d_lp.p + (size_t)3 * i, gp.h_lp.p + (size_t)3 * i, 24);
            }
            set_small_attr();
            const Patch *hdr = PL.finish();
            IncPrologue &pro = c.pro;
            pro.patches = hdr; pro.payload = (const char *)PL.buf.p; pro.n_patch = (int)PL.hdr.size(); pro.f_begin = Fold; pro.f_end = F;
            pro.fa = gp.d_fa.p; pro.fb = gp.d_fb.p; pro.Z = gp.d_z.p; pro.Wm = gp.d_W.p; pro.lp = gp.d_lp.p; pro.st = gp.d_state.p; pro.swp = c.d_swap.p;
            pro.slot_blk = c.dp.slot_blk; pro.slot_rhs = c.dp.slot_rhs; pro.Hc = c.d_H.p; pro.bad = c.d_bad.p; pro.stamps = nullptr; pro.done = nullptr; pro.seq = 0;
            pro.inl = g_opt.inc_inline && PL.hdr.size() <= (size_t)INL_PATCHES && PL.used <= (size_t)INL_BYTES;
            if (pro.inl) { memcpy(c.inl.hdr, PL.hdr.data(), PL.hdr.size() * sizeof(Patch)); memcpy(c.inl.pay, PL.buf.p, PL.used); }
            c.one_wait = 0;
            if (g_incprof_stamps) { c.h_kstamp.need(8 + PROF_SLOTS); memset(c.h_kstamp.p, 0, 8 * (8 + PROF_SLOTS)); pro.stamps = c.h_kstamp.p; }
            if (g_opt.inc_one_spin) {
                if (!c.h_done.p) { c.h_done.need(16); c.h_done.p[0] = 0; }
                c.done_seq = c.done_seq >= 0x7ffffff0 ? 1 : c.done_seq + 1;
                pro.done = c.h_done.p; pro.seq = c.done_seq; c.one_wait = c.done_seq;
            }
            gp.h_out.need((size_t)3 * N);
            const UpdArgs upd1{ c.d_perm.p, gp.d_lp.p, nullptr, gp.d_dx.p, gp.h_out.p, gp.h_dx.p, c.h_bad.p };
            const IncFlags nofl{ nullptr, 0, nullptr, 0, nullptr, 0 };
            const int *dn = c.d_tab.p + I.tab_used; const int n_dn = solve_here ? 0 : (int)lst.size();
            const int one_nt = g_opt.inc_one_threads >= 1024 ? 1024 : g_opt.inc_one_threads >= 512 ? 512 : 256;
            if (one_nt >= 1024) hipLaunchKernelGGL(k_inc_one<1024>, dim3(1), dim3(1024), lds, s, pro, nofl, c.inl, c.dp, tstep, (const int *)nullptr, 0, dn, n_dn, c.d_pool.p, 0ll, g_opt.block_factor, c.d_x.p, upd1);
            else if (one_nt >= 512) hipLaunchKernelGGL(k_inc_one<512>, dim3(1), dim3(512), lds, s, pro, nofl, c.inl, c.dp, tstep, (const int *)nullptr, 0, dn, n_dn, c.d_pool.p, 0ll, g_opt.block_factor, c.d_x.p, upd1);
            else hipLaunchKernelGGL(k_inc_one<256>, dim3(1), dim3(256), lds, s, pro, nofl, c.inl, c.dp, tstep, (const int *)nullptr, 0, dn, n_dn, c.d_pool.p, 0ll, g_opt.block_factor, c.d_x.p, upd1);
            gp.mirror_sync = true;
            gp.new_states = gp.h_out.p;
            HIPCHECK(hipGetLastError());
            c.pat.resize((size_t)2 * F);
            for (int f = Fold; f < F; f++) { c.pat[2 * f] = fa[f]; c.pat[2 * f + 1] = fb[f]; }
            c.patN = N;
            c.st.n_fronts = nFr0; c.st.n_levels = I.nLev0 + nT0;
            return true;
        }
        tstep.solve = 0; tstep.s_idx = 0;                 // (the general path below runs the back substitution in launches of its own)
    }
    // (any other way of factorising that front reads its destination records: they are brought up to date first)
    if (!tail_fast && I.recs_stale >= 0) I.dirty[I.recs_stale] = 1;
    // ---- 0. new poses join the last tail front, or open the next one --------------------------------------------------
    auto n_tail = [&]() { return (int)I.t_first.size(); };
    int grown_lo = 1 << 30;                              // first tail front that gained own poses in this step (several may: a front fills up, the next opens)
    for (int k = std::max(Nold, Nb); k < N; k++) {
        if (I.t_first.empty() || I.t_cnt.back() >= TAIL_POSES) {
            if (n_tail() >= MAX_TAIL_FRONTS - 1) return inc_fail(4);
            I.t_first.push_back(k); I.t_cnt.push_back(0);
            const int t = nF0 + n_tail() - 1;
            I.parent.push_back(-1); I.E.emplace_back(); I.xfac.emplace_back(); I.rel_begin.push_back(0); I.cur_nub.push_back(0); I.cur_cap.push_back(0);
            I.dirty.push_back(0); I.f_level.push_back(I.nLev0 + n_tail() - 1); I.kids.emplace_back(); I.stale.push_back(0);
            I.fd.emplace_back(); memset(&I.fd[t], 0, sizeof(FrontDesc));
            I.fd[t].first = k; I.fd[t].parent = -1; I.fd[t].dinv0 = -1;
        }
        I.t_cnt.back()++;
        I.tf_of.push_back(nF0 + n_tail() - 1);
        I.dirty[nF0 + n_tail() - 1] = 1;                 // its own columns changed
        grown_lo = std::min(grown_lo, nF0 + n_tail() - 1);
    }
    const int nT = n_tail(), nFr = nF0 + nT;
    auto is_tail = [&](int t) { return t >= nF0; };
    auto owns = [&](int t, int node) {                 // node (a TAIL pose) among the own columns of front t?
        return is_tail(t) && node >= I.t_first[t - nF0] && node < I.t_first[t - nF0] + I.t_cnt[t - nF0];
    };
    auto set_parent = [&](int t, int par) {
        if (I.parent[t] == par) return;
        if (I.parent[t] >= nF0) { auto &kd = I.kids[I.parent[t]]; kd.erase(std::find(kd.begin(), kd.end(), t)); }
        I.parent[t] = par;
        if (par >= nF0) { auto &kd = I.kids[par]; kd.insert(std::lower_bound(kd.begin(), kd.end(), t), t); }
    };
    // tail pose k enters the structure of front t and of every front above it, up to the front that owns k.  Fronts
    // without a base parent hang below the tail front that owns the first tail pose of their structure.
    bool unfit = false;
    std::vector<char> &mid = I.st_mid; mid.assign(nFr, 0);      // fronts whose structure gained a row that is NOT its last one (their arrays cannot be updated by appending)
    auto add_struct = [&](int t, int k) {
        while (t >= 0 && !owns(t, k)) {
            auto &E = I.E[t];
            auto it = std::lower_bound(E.begin(), E.end(), k);
            if (it == E.end() || *it != k) { if (it != E.end()) mid[t] = 1; E.insert(it, k); I.dirty[t] = 1; }
            if (is_tail(t) || P.f_parent[t] < 0) {
                const int par = I.tf_of[E.front() - Nb];
                if (I.parent[t] >= 0 && I.parent[t] != par) { unfit = true; return; }     // re-parenting a front with structure: re-plan
                set_parent(t, par);
            }
            t = I.parent[t];
        }
    };
    const double tsub0 = now_ms();
    // ---- 1. owners of the new factors, tail rows along root paths ----------------------------------------------
    std::vector<int> &owner_of = I.st_owner; owner_of.assign(F - Fold, -1);
    for (int f = Fold; f < F; f++) {
        const int a = fa[f], b = fb[f];
        const bool ta = a >= Nb, tb = b >= Nb;
        int owner;
        if (b < 0) owner = ta ? I.tf_of[a - Nb] : I.pos_front[P.pos[a]];
        else if (ta && tb) { owner = I.tf_of[std::min(a, b) - Nb]; add_struct(owner, std::max(a, b)); }
        else if (ta != tb) {
            const int j = ta ? b : a, k = ta ? a : b;
            owner = I.pos_front[P.pos[j]];
            add_struct(owner, k);
        } else {
            const int pa = P.pos[a], pb = P.pos[b];
            owner = I.pos_front[std::min(pa, pb)];
            if (local_base(owner, std::max(pa, pb)) < 0) return inc_fail(5);       // would change the frozen structure
        }
        if (unfit) return inc_fail(6);
        I.xfac[owner].push_back(f);
        I.dirty[owner] = 1;
        owner_of[f - Fold] = owner;
    }
    // (a batch step on the extended plan re-assembles EVERY front from its records: the ones update steps bypassed are rebuilt first)
    if (batch) for (int t = 0; t < nFr; t++) if (I.stale[t]) I.dirty[t] = 1;
    for (int t = 0; t < nFr; t++) if (I.dirty[t] && I.parent[t] >= 0) I.dirty[I.parent[t]] = 1;     // (parents have larger ids)
    // ---- 1b. which dirty fronts take a low-rank UPDATE of their factor instead of being re-assembled and re-factorised ----------
    // (front_update_body).  Eligible: the front keeps its own columns (every front but the last tail front), the rows its
    // structure gained come last, its array is a single-workgroup one, the new factors it owns have a symmetric positive
    // definite W, and every dirty child is updated too (the vectors a front receives come from its children's updates).
    std::vector<char> &mode = I.st_mode; mode.assign(nFr, 0);
    std::vector<int> &fmask = I.st_mask, &fslot = I.st_slot; fmask.assign(nFr, 0); fslot.assign(F - Fold, -1);
    auto kids_of = [&](int t, const int **kb, const int **ke) {
        if (t >= nF0) { *kb = I.kids[t].data(); *ke = *kb + I.kids[t].size(); }
        else { *kb = P.ch_idx.data() + P.ch_ptr[t]; *ke = P.ch_idx.data() + P.ch_ptr[t + 1]; }
    };
    bool any_upd = false;
    int n_dirty = 0;
    bool all_small = true;                             // every dirty front still fits the single-workgroup kernel
    {
        const size_t small_max = (size_t)g_opt.small_lds_kb * 1024;
        const int nw = waves_of(small_threads_for(1));
        for (int t = 0; t < nFr; t++) {
            if (!I.dirty[t]) continue;
            n_dirty++;
            const int nsb = t >= nF0 ? I.t_cnt[t - nF0] : P.f_nsb[t];
            int nub = (t >= nF0 ? 0 : P.f_nub[t]) + (int)I.E[t].size();
            if (t == nFr - 1 && g_opt.inc_tail && I.E[t].empty()) nub += std::max(0, TAIL_POSES - nsb);      // (phantom rows of the last tail front)
            const int R = 3 * (nsb + nub + 1);
            all_small = all_small && (small_front_lds(R, R - 3, nw) <= small_max || (g_opt.panel_mode && panel_front_lds(R, 3 * nsb, nw) <= small_max));
        }
    }
    // A plan made of single-workgroup fronts only, one of which has collected so many rows of loop closures since that it no longer
    // fits the LDS: from here on every step on its root path would take the multi-launch big-front path (and no low-rank
    // updates).  The structure has outgrown the plan -- a fresh one is cheaper than what follows (measured on the M3500 demo:
    // 13 such steps, re-planned 505 ms in total, carried on 590 ms).
    if (!batch && !all_small && !I.base_has_big && g_opt.inc_replan_tall) return inc_fail(17);
    if (!batch && g_opt.inc_update && I.upd_ok && !tail_fast && g_opt.persist && g_opt.inc_multi && g_opt.wave_backsolve) {
        const size_t small_max = (size_t)g_opt.small_lds_kb * 1024;
        const int nw = waves_of(small_threads_for(1));
        int n_slots = 0;
        for (int t = 0; t < nFr && all_small && n_dirty <= g_opt.persist_max_fronts; t++) {
            if (!I.dirty[t] || t == nFr - 1 || t >= grown_lo || mid[t] || I.cur_cap[t] <= 0) continue;      // (t >= grown_lo: its own columns changed)
            const int nsb = t >= nF0 ? I.t_cnt[t - nF0] : P.f_nsb[t], nub = (t >= nF0 ? 0 : P.f_nub[t]) + (int)I.E[t].size();
            const int R = 3 * (nsb + nub + 1);
            bool ok = small_front_lds(R, R - 3, nw) <= small_max || (g_opt.panel_mode && panel_front_lds(R, 3 * nsb, nw) <= small_max);
            const int *kb, *ke; kids_of(t, &kb, &ke);
            int ndc = 0, msk = 0;
            for (const int *kp = kb; kp != ke && ok; kp++) if (I.dirty[*kp]) { ok = mode[*kp] != 0; msk |= fmask[*kp]; ndc++; }
            ok = ok && ndc <= UPD_MAXC;
            int nown = 0;
            for (size_t q = I.xfac[t].size(); q-- > 0 && ok;) {
                const int f = I.xfac[t][q];
                if (f < Fold) break;                   // (the factors of this step are the last ones of the list)
                // W = C C^T must exist: symmetric, pivots well away from zero (the kernel repeats this factorisation)
                const double *w = gp.h_W.p + (size_t)9 * f;
                ok = w[1] == w[3] && w[2] == w[6] && w[5] == w[7] && w[0] > 0;
                if (ok) {
                    const double c00 = std::sqrt(w[0]), c10 = w[3] / c00, c20 = w[6] / c00, d1 = w[4] - c10 * c10;
                    ok = d1 > 1e-12 * w[4];
                    if (ok) { const double c11 = std::sqrt(d1), c21 = (w[7] - c20 * c10) / c11, d2 = w[8] - c20 * c20 - c21 * c21; ok = d2 > 1e-12 * w[8]; }
                }
                nown++;
            }
            ok = ok && nown <= UPD_MAXF && n_slots + nown <= UPD_MAXF;
            if (ok) {
                for (size_t q = I.xfac[t].size(); q-- > 0;) { const int f = I.xfac[t][q]; if (f < Fold) break; fslot[f - Fold] = n_slots; msk |= 1 << n_slots; n_slots++; }
                ok = msk != 0 && update_front_lds(R, 3 * nsb, __builtin_popcount(msk)) <= (size_t)160 * 1024;
            }
            if (ok) { mode[t] = 1; fmask[t] = msk; any_upd = true; }
        }
    }
    const double tsub1 = now_ms();
    // ---- 2. regenerate dirty fronts (children before parents) ----------------------------------------------------
    std::vector<int> &st_i32 = I.st_i32; std::vector<DestRec> &st_dest = I.st_dest; std::vector<ChildRec> &st_child = I.st_child;
    st_i32.clear(); st_dest.clear(); st_child.clear();
    const int nLev = I.nLev0 + nT;
    std::vector<std::vector<int>> lev_dirty(nLev);
    std::vector<int> fd_dirty;
    std::vector<int> &new_slot_blk = I.st_sb, &new_slot_rhs = I.st_sr; std::vector<unsigned char> &new_swap = I.st_sw;
    new_slot_blk.assign((size_t)3 * (F - Fold), -1); new_slot_rhs.assign((size_t)2 * (F - Fold), -1); new_swap.assign(F - Fold, 0);
    c.inc_slot_blk.resize((size_t)3 * (F - I.Fb), -1); c.inc_slot_rhs.resize((size_t)2 * (F - I.Fb), -1);
    for (int f = Fold; f < F; f++) {                     // 5 fresh slots per new factor (3 blocks, 2 rhs segments)
        for (int k = 0; k < 3; k++) new_slot_blk[(size_t)3 * (f - Fold) + k] = c.inc_slot_blk[(size_t)3 * (f - I.Fb) + k] = I.slots_used++;
        for (int k = 0; k < 2; k++) new_slot_rhs[(size_t)2 * (f - Fold) + k] = c.inc_slot_rhs[(size_t)2 * (f - I.Fb) + k] = I.slots_used++;
    }
    const long long i32_base = I.i32_used, dest_base = I.dest_used, child_base = I.child_used;
    std::vector<UpdRec> &rec_of = I.st_rec; std::vector<int> &wout_of = I.st_wout;
    if (any_upd) { rec_of.resize(nFr); wout_of.assign(nFr, 0); }
    long long wbuf_used = 0;
    struct Ent { int col, row, f, k, slot; };
    std::vector<Ent> ents;
    auto nsb_of = [&](int t) { return is_tail(t) ? I.t_cnt[t - nF0] : P.f_nsb[t]; };
    auto nub0_of = [&](int t) { return is_tail(t) ? 0 : P.f_nub[t]; };
    for (int t = 0; t < nFr; t++) {
        if (!I.dirty[t]) continue;
        const bool tail = is_tail(t);
        const int nsb = nsb_of(t), nub0 = nub0_of(t);
        const std::vector<int> &E = I.E[t];
        // the last tail front keeps the shape of a FULL one while it fills up: phantom structure rows (zero rows of L, x taken
        // from a position that stays zero) stand in for the poses still to come, so that its leading dimension and the place of
        // its right-hand-side row do not move when a pose arrives -- what tail_refactor relies on
        const int nph = (tail && t == nFr - 1 && g_opt.inc_tail && E.empty()) ? std::max(0, TAIL_POSES - nsb) : 0;
        const int nub = nub0 + (int)E.size() + nph, nbc = nsb + nub;
        const long long need = (long long)(3 * (nbc + 1)) * (3 * nbc);
        FrontDesc &D = I.fd[t];
        if (tail_fast && t == tstep.t) {               // only the descriptor changes: records, children and array stay
            D.nsb = nsb; D.nub = nub; I.cur_nub[t] = nub;
            for (int f = Fold; f < F; f++) new_swap[f - Fold] = fb[f] >= 0 && fa[f] < fb[f];      // (own poses: local order = id order; see add_factor below)
            fd_dirty.push_back(t);
            I.recs_stale = t;
            continue;
        }
        if (t == I.recs_stale) I.recs_stale = -1;
        const bool upd = any_upd && mode[t];
        const int old_nub = I.cur_nub[t]; const long long old_off = D.off;
        // (an updated front whose structure grew is written to a FRESH array in the new layout -- nothing moves in place; one that
        // keeps its structure is updated where it is)
        if (upd ? nub != old_nub : need > I.cur_cap[t]) {
            // growing fronts (the last tail front, fronts collecting tail rows) get head-room: no new array every step
            const int gb = upd ? nbc : tail ? std::max(nbc + 4, TAIL_POSES + (int)E.size() + 4) : nbc + 4;
            const long long want = (long long)(3 * (gb + 1)) * (3 * gb);
            const long long off = (I.pool_used + 31) & ~31ll;
            if (off + want > I.pool_cap) return inc_fail(7);
            D.off = off; I.pool_used = off + want; I.cur_cap[t] = want;
        }
        D.nsb = nsb; D.nub = nub; I.cur_nub[t] = nub;
        if (tail) D.first = I.t_first[t - nF0];
        // struct rows (positions): base struct then tail nodes (position of a tail node = its id)
        D.rows_begin = (int)(i32_base + (long long)st_i32.size());
        if (!tail) st_i32.insert(st_i32.end(), P.f_rows.begin() + P.f_rows_ptr[t], P.f_rows.begin() + P.f_rows_ptr[t + 1]);
        st_i32.insert(st_i32.end(), E.begin(), E.end());
        if (nph > 0) st_i32.insert(st_i32.end(), (size_t)TAIL_POSES, I.zpos);       // (phantom rows; a full run: the descriptor's nub shrinks as the front fills up)
        auto local = [&](int node) -> int {            // local block index of a node in this front
            if (node >= Nb) {
                if (tail && node < D.first + nsb) return node - D.first;
                auto it = std::lower_bound(E.begin(), E.end(), node);
                return nsb + nub0 + (int)(it - E.begin());
            }
            return local_base(t, P.pos[node]);
        };
        if (upd) {
            // low-rank update: no destination records, no child records (stale from here on: rebuilt when the front is next
            // re-assembled); what the kernel needs is where the vectors come from and where its own go
            UpdRec u; memset(&u, 0, sizeof(u));
            u.old_off = old_off; u.mode = 1; u.old_nub = old_nub; u.mask = fmask[t];
            u.wout = (int)wbuf_used; wout_of[t] = u.wout;
            wbuf_used += (long long)3 * UPD_MAXF * (3 * nub + 1);
            if (wbuf_used > (long long)c.d_wbuf.cap) return inc_fail(8);
            for (int f : I.xfac[t]) {
                if (f < Fold) continue;
                const int la = local(fa[f]), lb = fb[f] >= 0 ? local(fb[f]) : -1;
                new_swap[f - Fold] = (lb >= 0 && la < lb);         // (orientation of the off-diagonal block in its contribution slot, for later re-assemblies)
                u.own_f[u.n_own] = f; u.own_la[u.n_own] = la; u.own_lb[u.n_own] = lb; u.own_slot[u.n_own] = fslot[f - Fold]; u.n_own++;
            }
            const int *kb, *ke; kids_of(t, &kb, &ke);
            for (const int *kp = kb; kp != ke; kp++) {
                const int ch = *kp;
                if (!I.dirty[ch]) continue;
                const std::vector<int> &Ec = I.E[ch];
                I.rel_begin[ch] = (int)(i32_base + (long long)st_i32.size());
                if (ch < nF0) st_i32.insert(st_i32.end(), P.f_rel.begin() + P.f_rows_ptr[ch], P.f_rel.begin() + P.f_rows_ptr[ch + 1]);
                for (int k : Ec) st_i32.push_back(local(k));
                u.ch_t[u.n_ch] = ch; u.ch_wout[u.n_ch] = wout_of[ch]; u.ch_rel[u.n_ch] = I.rel_begin[ch];
                u.ch_cnu[u.n_ch] = nub0_of(ch) + (int)Ec.size(); u.ch_mask[u.n_ch] = fmask[ch]; u.n_ch++;
            }
            rec_of[t] = u;
            I.stale[t] = 1;
            D.parent = I.parent[t];
            fd_dirty.push_back(t);
            lev_dirty[I.f_level[t]].push_back(t);
            continue;
        }
        I.stale[t] = 0;
        // destination records: only fronts that own factors added since the batch need new ones
        if (!I.xfac[t].empty()) {
            ents.clear();
            auto add_factor = [&](int f) {
                const int a = fa[f], b = fb[f];
                const int la = local(a), lb = b >= 0 ? local(b) : -1;
                if (f >= Fold) new_swap[f - Fold] = (lb >= 0 && la < lb);       // orientation of the off-diagonal block
                const int *sb = f < I.Fb ? &P.slot_blk[(size_t)3 * f] : &c.inc_slot_blk[(size_t)3 * (f - I.Fb)];
                const int *sr = f < I.Fb ? &P.slot_rhs[(size_t)2 * f] : &c.inc_slot_rhs[(size_t)2 * (f - I.Fb)];
                ents.push_back({ la, la, f, 0, sb[0] }); ents.push_back({ la, -1, f, 3, sr[0] });
                if (lb >= 0) {
                    ents.push_back({ std::min(la, lb), std::max(la, lb), f, 1, sb[1] });
                    ents.push_back({ lb, lb, f, 2, sb[2] }); ents.push_back({ lb, -1, f, 4, sr[1] });
                }
            };
            if (!tail) for (int q = I.bf_ptr[t]; q < I.bf_ptr[t + 1]; q++) add_factor(I.bf_idx[q]);
            for (int f : I.xfac[t]) add_factor(f);
            std::sort(ents.begin(), ents.end(), [](const Ent &x, const Ent &y) {
                if (x.col != y.col) return x.col < y.col;
                if (x.row != y.row) return x.row < y.row;
                if (x.f != y.f) return x.f < y.f;
                return x.k < y.k;
            });
            D.dest_begin = (int)(dest_base + (long long)st_dest.size());
            for (size_t i = 0; i < ents.size(); i++) {
                const bool fresh = i == 0 || ents[i].col != ents[i - 1].col || ents[i].row != ents[i - 1].row;
                if (fresh) st_dest.push_back({ ents[i].row, ents[i].col, (int)(i32_base + (long long)st_i32.size()), 0 });
                st_i32.push_back(ents[i].slot);
                st_dest.back().src_end = -(int)(i32_base + (long long)st_i32.size());
            }
            D.dest_end = (int)(dest_base + (long long)st_dest.size());
        }
        // children: records + block maps into this front's (possibly longer) row list
        const int *kb, *ke;
        if (tail) { kb = I.kids[t].data(); ke = kb + I.kids[t].size(); }
        else { kb = P.ch_idx.data() + P.ch_ptr[t]; ke = P.ch_idx.data() + P.ch_ptr[t + 1]; }
        D.ch_begin = (int)(child_base + (long long)st_child.size());
        for (const int *kp = kb; kp != ke; kp++) {
            const int ch = *kp;
            const int cnsb = nsb_of(ch), cnub0 = nub0_of(ch);
            const std::vector<int> &Ec = I.E[ch];
            I.rel_begin[ch] = (int)(i32_base + (long long)st_i32.size());
            if (!tail) st_i32.insert(st_i32.end(), P.f_rel.begin() + P.f_rows_ptr[ch], P.f_rel.begin() + P.f_rows_ptr[ch + 1]);
            for (int k : Ec) st_i32.push_back(local(k));
            ChildRec r;
            r.cnu = cnub0 + (int)Ec.size(); r.cR = 3 * (cnsb + r.cnu + 1);
            r.uoff = I.fd[ch].off + (long long)(3 * cnsb) * r.cR + 3 * cnsb;
            r.rel_begin = I.rel_begin[ch]; r.pad = ch;
            st_child.push_back(r);
        }
        D.ch_end = (int)(child_base + (long long)st_child.size());
        D.parent = I.parent[t];
        fd_dirty.push_back(t);
        lev_dirty[I.f_level[t]].push_back(t);
    }
    if (I.i32_used + (long long)st_i32.size() > (long long)c.d_i32.cap || I.dest_used + (long long)st_dest.size() > (long long)c.d_dest.cap ||
        I.child_used + (long long)st_child.size() > (long long)c.d_child.cap || (size_t)9 * I.slots_used > c.d_H.cap || (size_t)nFr > c.d_fd.cap ||
        (size_t)N > c.d_perm.cap) return inc_fail(9);
    const double tsub2 = now_ms();
    // ---- 3. launch tables of the dirty fronts (transient region behind the base tables) + back-substitution lists -----
    std::vector<int> &tab = I.st_tab; tab.clear(); std::vector<LevelPlan> dl(nLev);
    auto dims = [&](int t, int *nsb, int *nub) { *nsb = nsb_of(t); *nub = I.cur_nub[t]; };
    if (batch) {            // every front of every level, with its current dimensions
        for (int l = 0; l < nLev; l++) {
            if (l < I.nLev0) lev_dirty[l].assign(P.lev_fronts.begin() + P.lev_ptr[l], P.lev_fronts.begin() + P.lev_ptr[l + 1]);
            else lev_dirty[l].assign(1, nF0 + l - I.nLev0);
        }
    }
    for (int l = 0; l < nLev; l++) if (!lev_dirty[l].empty()) build_level(dl[l], lev_dirty[l], tab, dims, [&](int t) { return I.fd[t].dinv0 >= 0; });
    // back substitution: tail fronts one by one (last first), then the base levels; restricted to the fronts that hold a
    // visited pose (and their ancestors) when the reference only walks the marked root paths
    I.need.assign(nFr, needed ? 0 : 1);
    if (needed && !batch) {
        for (const RefModel::Visit &v : *needed) {
            int t = v.node >= Nb ? I.tf_of[v.node - Nb] : I.pos_front[P.pos[v.node]];
            while (t >= 0 && !I.need[t]) { I.need[t] = 1; t = I.parent[t]; }
        }
    }
    std::vector<int> bs_off(nLev, -1), bs_n(nLev, 0), bs_maxns(nLev, 0);
    std::vector<size_t> bs_wlds(nLev, 0);
    auto bs_note = [&](int l, int t) { bs_wlds[l] = std::max(bs_wlds[l], backsolve_lds(3 * (nsb_of(t) + I.cur_nub[t]), 3 * nsb_of(t), true)); bs_maxns[l] = std::max(bs_maxns[l], 3 * nsb_of(t)); };
    for (int l = nLev - 1; l >= 0; l--) {
        bs_off[l] = (int)(I.tab_used + (long long)tab.size());
        if (batch) continue;                       // (the full tables built above serve the back substitution too)
        if (l >= I.nLev0) { const int t = nF0 + (l - I.nLev0); if (I.need[t]) { tab.push_back(t); bs_n[l] = 1; bs_note(l, t); } }
        else if (needed) {
            for (int k = I.base_levels[l].all_off; k < I.base_levels[l].all_off + I.base_levels[l].n_all; k++) {
                const int t = c.base_tab[k];
                if (I.need[t]) { tab.push_back(t); bs_n[l]++; bs_note(l, t); }
            }
        }
    }
    if (I.tab_used + (long long)tab.size() > (long long)c.d_tab.cap) return inc_fail(10);
    for (int l = 0; l < nLev; l++) {
        if (lev_dirty[l].empty()) continue;
        LevelPlan &L = dl[l];
        const int sh = (int)I.tab_used;
        L.all_off += sh; L.small_off += sh; L.asm_big.list_off += sh; L.asm_big.pre_off += sh; L.asm_tile.list_off += sh; L.asm_tile.pre_off += sh;
        for (auto &x : L.panel) { x.list_off += sh; x.pre_off += sh; }
        for (auto &x : L.syrk) { x.list_off += sh; x.pre_off += sh; }
        for (auto &x : L.syrkw) { x.list_off += sh; x.pre_off += sh; }
        for (auto &x : L.syrka) { x.list_off += sh; x.pre_off += sh; }
        for (auto &x : L.syrkb) { x.list_off += sh; x.pre_off += sh; }
        for (auto &x : L.bchain) { x.list_off += sh; x.pre_off += sh; }
        for (auto &x : L.btile) { x.list_off += sh; x.pre_off += sh; }
        L.bs_blk.list_off += sh; L.bs_blk.pre_off += sh; L.rest_off += sh;
        L.bs_gemv.list_off += sh; L.bs_gemv.pre_off += sh; L.wb_off += sh;
        if (l < I.nLev0) for (int t : lev_dirty[l]) {
            I.base_levels[l].solve_lds = std::max(I.base_levels[l].solve_lds, (size_t)(3 * (P.f_nsb[t] + I.cur_nub[t]) + NB + 8 + NB * (NB + 1)) * 8);
            I.base_levels[l].solve_w_lds = std::max(I.base_levels[l].solve_w_lds, backsolve_lds(3 * (P.f_nsb[t] + I.cur_nub[t]), 3 * P.f_nsb[t], true));
        }
    }
    for (int l = 0; l < nLev; l++) if (diag_doubles(dl[l].n_big, dl[l].n_diag_slots) > c.d_diag.cap) return inc_fail(11);
    // batch on the extended plan: levels >= 1 as ONE multi-level launch per sweep (see enqueue_numeric), when they hold small
    // fronts only
    int mp_up_off = 0, mp_dn_off = 0, mp_n = 0, mp_nt = 0; size_t mp_up_lds = 0, mp_dn_lds = 0; long long mp_full = 0; int mp_dn_maxns = 0;
    bool mp = batch && g_opt.persist && nLev >= 3;
    if (mp) {
        mp_nt = dl[nLev - 1].small_nt;
        for (int l = 1; l < nLev && mp; l++) {
            const LevelPlan &L = dl[l];
            mp = L.n_big == 0 && L.bs_gemv.grid == 0 && L.small_nt == mp_nt && L.n_small == L.n_all;
            mp_n += L.n_all;
        }
        mp = mp && mp_n <= g_opt.persist_max_fronts && (size_t)2 * nFr <= c.d_flags.cap;
    }
    if (mp) {
        const int sh = (int)I.tab_used;
        const size_t tab_size0 = tab.size();
        mp_up_off = sh + (int)tab.size();
        for (int l = 1; l < nLev; l++) {
            const LevelPlan &L = dl[l];
            for (int k = 0; k < L.n_small; k++) tab.push_back(tab[L.small_off - sh + k]);
            mp_up_lds = std::max(mp_up_lds, L.small_lds); mp_full = std::max(mp_full, L.full_limit);
            for (int k = 0; k < L.n_all; k++) { const int t = tab[L.all_off - sh + k]; mp_dn_lds = std::max(mp_dn_lds, backsolve_lds(3 * (nsb_of(t) + I.cur_nub[t]), 3 * nsb_of(t), true)); mp_dn_maxns = std::max(mp_dn_maxns, 3 * nsb_of(t)); }
        }
        mp_dn_off = sh + (int)tab.size();
        for (int l = nLev - 1; l >= 1; l--) { const LevelPlan &L = dl[l]; for (int k = 0; k < L.n_all; k++) tab.push_back(tab[L.all_off - sh + k]); }
        if (I.tab_used + (long long)tab.size() > (long long)c.d_tab.cap || mp_dn_lds > 160 * 1024) { mp = false; tab.resize(tab_size0); }
    }
    auto solve_lds_of = [&](int t) { return (size_t)(3 * (nsb_of(t) + I.cur_nub[t]) + NB + 8 + NB * (NB + 1)) * 8; };
    for (int t = 0; t < nFr; t++) if (I.need[t] && solve_lds_of(t) > 160 * 1024) return inc_fail(12);
    // incremental step: the regenerated fronts of ALL levels as one multi-level launch (dependency flags, as the batch sweeps
    // do over the top of the tree), and the back substitution from the top as another -- a step that touches a root path is
    // three launches (prologue, fronts, back substitution + state update) instead of one per level and direction
    int iu_off = 0, iu_n = 0, iu_nt = 0; size_t iu_lds = 0; long long iu_full = 0;
    int id_off = 0, id_n = 0, id_maxns = 0, id_rest = -1; size_t id_lds = 0;      // id_rest: first level (downwards) left to per-level launches
    bool iu = !batch && g_opt.persist && g_opt.inc_multi, id = iu;
    if (iu) {
        const int sh = (int)I.tab_used;
        for (int l = 0; l < nLev && iu; l++) {
            if (lev_dirty[l].empty()) continue;
            const LevelPlan &L = dl[l];
            if (iu_nt == 0) iu_nt = L.small_nt;
            iu = L.n_big == 0 && L.n_small == L.n_all && L.small_nt == iu_nt;
            iu_n += L.n_all; iu_lds = std::max(iu_lds, L.small_lds); iu_full = std::max(iu_full, L.full_limit);
        }
        iu = iu && (iu_n >= 1 || tail_fast) && iu_n <= g_opt.persist_max_fronts;
        if (iu) {
            iu_off = sh + (int)tab.size();
            for (int l = 0; l < nLev; l++) if (!lev_dirty[l].empty()) for (int k = 0; k < dl[l].n_small; k++) tab.push_back(tab[dl[l].small_off - sh + k]);
        }
    }
    // updated fronts only exist inside that launch (or k_inc_one's loop over the same list): one record per list entry
    if (any_upd) {
        if (!iu || (size_t)iu_n > c.d_upd.cap || (size_t)3 * nFr > c.d_flags.cap) return inc_fail(13);      // (the eligibility pass checked what iu checks: a full re-plan otherwise)
        I.st_upd.resize(iu_n);
        const int sh = (int)I.tab_used;
        for (int i = 0; i < iu_n; i++) {
            const int t = tab[iu_off - sh + i];
            if (mode[t]) {
                I.st_upd[i] = rec_of[t];
                iu_lds = std::max(iu_lds, update_front_lds(3 * (nsb_of(t) + I.cur_nub[t] + 1), 3 * nsb_of(t), __builtin_popcount(fmask[t])));
            } else memset(&I.st_upd[i], 0, sizeof(UpdRec));
        }
    }
    if (id) {
        if (needed) {                                   // the lists of the marked root paths are contiguous, top level first
            id_off = bs_off[nLev - 1];
            for (int l = nLev - 1; l >= 0; l--) { id_n += bs_n[l]; id_lds = std::max(id_lds, bs_wlds[l]); id_maxns = std::max(id_maxns, bs_maxns[l]); }
            id = id_n >= 1 && id_n <= g_opt.persist_max_fronts && id_lds <= 160 * 1024;
        } else {                                        // every pose: the tail fronts and as many base levels as may be resident together
            id_off = (int)(I.tab_used + (long long)tab.size());
            id_rest = nLev - 1;
            for (int l = nLev - 1; l >= 0; l--) {
                size_t lds = 0; int mx = 0, n = 0;
                if (l >= I.nLev0) { const int t = nF0 + (l - I.nLev0); lds = backsolve_lds(3 * (nsb_of(t) + I.cur_nub[t]), 3 * nsb_of(t), true); mx = 3 * nsb_of(t); n = 1; }
                else { const LevelPlan &L = I.base_levels[l]; lds = L.solve_w_lds; mx = L.maxns; n = L.n_all; }
                if (id_n + n > g_opt.persist_max_fronts || std::max(id_lds, lds) > 160 * 1024) break;
                if (l >= I.nLev0) tab.push_back(nF0 + (l - I.nLev0));
                else for (int k = 0; k < n; k++) tab.push_back(c.base_tab[I.base_levels[l].all_off + k]);
                id_n += n; id_lds = std::max(id_lds, lds); id_maxns = std::max(id_maxns, mx); id_rest = l - 1;
            }
            id = id_n >= 2;
            if (!id) id_rest = -1;
        }
    }
    // ... and a step that regenerates a front or three and walks a short root path runs as ONE launch of one workgroup
    // (k_inc_one: prologue, fronts, back substitution one after the other)
    bool one = iu && id && needed && g_opt.inc_one && iu_n <= g_opt.inc_one_up && id_n <= g_opt.inc_one_dn && id_maxns <= BSW_MAX_NS && g_opt.wave_backsolve;
    const int one_nt = g_opt.inc_one_threads >= 1024 ? 1024 : g_opt.inc_one_threads >= 512 ? 512 : 256;
    size_t one_lds = std::max(id_lds, tail_fast ? tail_refactor_lds() : (size_t)0);
    if (one) {
        for (int l = 0; l < nLev; l++) for (int t : lev_dirty[l]) {         // the kernel's own full / panel decision, at its thread count
            const int R = 3 * (nsb_of(t) + I.cur_nub[t] + 1), C = R - 3;
            if (any_upd && mode[t]) { one_lds = std::max(one_lds, update_front_lds(R, 3 * nsb_of(t), __builtin_popcount(fmask[t]))); continue; }
            const size_t full = small_front_lds(R, C, one_nt / 64);
            one_lds = std::max(one_lds, (long long)full <= iu_full ? full : panel_front_lds(R, 3 * nsb_of(t), one_nt / 64));
        }
        one = one_lds <= 160 * 1024;
    }
    if (!one) { iu = iu && (iu_n >= 2 || any_upd); id = id && id_n >= 2; }
    if (tail_fast && !one && tail_refactor_lds() > 64 * 1024) return inc_fail(14);       // (never: the refactorisation alone runs as k_inc_one without lists)
    if (I.tab_used + (long long)tab.size() > (long long)c.d_tab.cap) return inc_fail(15);
    c.st.reserved0 = (int)fd_dirty.size();              // fronts regenerated by this step (tools/inc_hist.py)
    if (!batch) {
        int nu_ = 0; for (int t : fd_dirty) nu_ += (any_upd && mode[t]) ? 1 : 0;
        c.st.inc_fronts_updated = nu_;
        if (any_upd) { g_updstat[0]++; g_updstat[2] += nu_; g_updstat[3] += (long long)fd_dirty.size() - nu_; } else { g_updstat[1]++; g_updstat[4] += (long long)fd_dirty.size(); }
        if (one) g_updstat[5]++;
    }
    const double tsub3 = now_ms();
    // ---- 4. uploads: every table update of this step, the new factors and the new states through ONE pinned staging
    //         buffer, scattered by one kernel (k_apply_patches) -- no copy-engine call on the path ---------------------------
    PatchList &PL = c.patches;
    PL.reset();
    {   // new factors (what upload_factors would copy)
        if (F > gp.F_cap) return inc_fail(16);                        // device arrays must grow: the re-plan path re-uploads
        const int f0 = gp.F_on_device;
        if (F > f0) {
            PL.add(gp.d_fa.p + f0, gp.h_fa.p + f0, (size_t)(F - f0) * 4); PL.add(gp.d_fb.p + f0, gp.h_fb.p + f0, (size_t)(F - f0) * 4);
            PL.add(gp.d_z.p + (size_t)3 * f0, gp.h_z.p + (size_t)3 * f0, (size_t)(F - f0) * 24);
            PL.add(gp.d_W.p + (size_t)9 * f0, gp.h_W.p + (size_t)9 * f0, (size_t)(F - f0) * 72);
            gp.F_on_device = F;
        }
    }
    PL.add(c.d_i32.p + I.i32_used, st_i32.data(), st_i32.size() * 4);
    PL.add(c.d_dest.p + I.dest_used, st_dest.data(), st_dest.size() * sizeof(DestRec));
    PL.add(c.d_child.p + I.child_used, st_child.data(), st_child.size() * sizeof(ChildRec));
    PL.add(c.d_tab.p + I.tab_used, tab.data(), tab.size() * 4);
    I.i32_used += (long long)st_i32.size(); I.dest_used += (long long)st_dest.size(); I.child_used += (long long)st_child.size();
    for (int t : fd_dirty) PL.add(c.d_fd.p + t, &I.fd[t], sizeof(FrontDesc));
    if (any_upd) PL.add(c.d_upd.p, I.st_upd.data(), I.st_upd.size() * sizeof(UpdRec));
    if (F > Fold) {
        PL.add((int *)c.dp.slot_blk + (size_t)3 * Fold, new_slot_blk.data(), new_slot_blk.size() * 4);
        PL.add((int *)c.dp.slot_rhs + (size_t)2 * Fold, new_slot_rhs.data(), new_slot_rhs.size() * 4);
        PL.add(c.d_swap.p + Fold, new_swap.data(), new_swap.size());
    }
    if (N > Nold) {
        std::vector<int> &ids = I.st_ids; std::vector<double> &zeros = I.st_zeros;
        ids.resize(N - Nold); zeros.assign(N - Nold, 0.0);
        for (int i = Nold; i < N; i++) ids[i - Nold] = i;
        PL.add(c.d_pos.p + Nold, ids.data(), ids.size() * 4);
        PL.add(c.d_perm.p + Nold, ids.data(), ids.size() * 4);          // (tail poses are eliminated in id order: position = id)
        if (!batch) PL.add(c.d_lambda.p + Nold, zeros.data(), zeros.size() * 8);      // no Tikhonov term on poses added incrementally (aprilsam.c:508-542)
        c.lambda_N = -1;
        P.perm.resize(N); P.pos.resize(N);
        for (int i = Nold; i < N; i++) { P.perm[i] = i; P.pos[i] = i; }
    }
    if (batch) {            // aprilsam.c:197-204: the Tikhonov term on every pose
        c.h_lambda.assign(N, batch_lambda);
        PL.add(c.d_lambda.p, c.h_lambda.data(), (size_t)N * 8);
        c.lambda_N = -1;
    }
    if (mp) {            // levels of the tail fronts for the dependency flags (base fronts: uploaded with the plan)
        I.st_ids.resize(nT);
        for (int i = 0; i < nT; i++) I.st_ids[i] = I.nLev0 + i;
        PL.add(c.d_flevel.p + nF0, I.st_ids.data(), (size_t)nT * 4);
    }
    const double tsub4 = now_ms();
    // ---- 5. numeric: new factors linearised, dirty fronts level by level, back substitution, update ----------------------
    set_small_attr();
    const UpdCtx uctx = any_upd ? UpdCtx{ c.d_upd.p, c.d_wbuf.p, c.d_flags.p + (size_t)2 * nFr, gp.d_fa.p, gp.d_fb.p, gp.d_z.p, gp.d_W.p, gp.d_lp.p, gp.d_state.p } : UpdCtx{};
    if (batch) {
        PL.launch(s);
        hipLaunchKernelGGL(k_load_states, dim3((3 * N + TPB - 1) / TPB), dim3(TPB), 0, s, 3 * N, gp.h_state.p, gp.d_state.p, gp.d_lp.p);      // l_point <- state
        hipLaunchKernelGGL((k_linearize_t<false>), dim3((F + TPB - 1) / TPB), dim3(TPB), 0, s, 0, F, (const int *)nullptr, gp.d_fa.p, gp.d_fb.p, gp.d_z.p, gp.d_W.p,
                           gp.d_lp.p, gp.d_state.p, c.d_swap.p, c.dp.slot_blk, c.dp.slot_rhs, c.d_H.p, c.d_bad.p, (const double *)nullptr,
                           nFr, c.d_flevel.p, 1, mp ? c.d_flags.p : (int *)nullptr);
    } else {
        // states first (the new factors are linearised at them; new priors at the node's current state) -- as patches of the few
        // poses whose host objects differ from the pinned mirror (pack_states_diff), or, when that is not known to be enough,
        // all of them from the mirrors -- then ONE single-workgroup launch for all patches + the linearisation of the new factors
        if (patch_states) {
            for (int i : gp.changed) {
                PL.add(gp.d_state.p + (size_t)3 * i, gp.h_state.p + (size_t)3 * i, 24);
                PL.add(gp.d_lp.p + (size_t)3 * i, gp.h_lp.p + (size_t)3 * i, 24);
            }
        } else {
            hipLaunchKernelGGL(k_load_states_lp, dim3((3 * N + TPB - 1) / TPB), dim3(TPB), 0, s, 3 * N, gp.h_state.p, gp.h_lp.p, gp.d_state.p, gp.d_lp.p, c.d_bad.p);
        }
        const Patch *hdr = PL.finish();
        IncPrologue &pro = c.pro;                      // (2.7 KB with the inline patch area: filled in place)
        pro.patches = hdr; pro.payload = (const char *)PL.buf.p; pro.n_patch = (int)PL.hdr.size(); pro.f_begin = Fold; pro.f_end = F;
        pro.fa = gp.d_fa.p; pro.fb = gp.d_fb.p; pro.Z = gp.d_z.p; pro.Wm = gp.d_W.p; pro.lp = gp.d_lp.p; pro.st = gp.d_state.p; pro.swp = c.d_swap.p;
        pro.slot_blk = c.dp.slot_blk; pro.slot_rhs = c.dp.slot_rhs; pro.Hc = c.d_H.p; pro.bad = c.d_bad.p; pro.stamps = nullptr; pro.done = nullptr; pro.seq = 0;
        pro.inl = g_opt.inc_inline && PL.hdr.size() <= (size_t)INL_PATCHES && PL.used <= (size_t)INL_BYTES;
        if (pro.inl) { memcpy(c.inl.hdr, PL.hdr.data(), PL.hdr.size() * sizeof(Patch)); memcpy(c.inl.pay, PL.buf.p, PL.used); }
        c.one_wait = 0;
        const IncFlags fl = (!one && (iu || id)) ? IncFlags{ c.d_flags.p, nFr, c.d_tab.p + iu_off, iu ? iu_n : 0, c.d_tab.p + id_off, id ? id_n : 0 } : IncFlags{ nullptr, 0, nullptr, 0, nullptr, 0 };
        if (tail_fast && !one) {                       // the refactorisation in the prologue's launch, the back substitution in launches of its own
            hipLaunchKernelGGL(k_inc_one<1024>, dim3(1), dim3(1024), tail_refactor_lds(), s, pro, fl, c.inl, c.dp, tstep, (const int *)nullptr, 0, (const int *)nullptr, 0,
                               c.d_pool.p, iu_full, g_opt.block_factor, c.d_x.p, UpdArgs{});
        } else if (one) {
            if (g_incprof_stamps) { c.h_kstamp.need(8 + PROF_SLOTS); memset(c.h_kstamp.p, 0, 8 * (8 + PROF_SLOTS)); pro.stamps = c.h_kstamp.p; }
            if (g_opt.inc_one_spin) {                  // completion through a word in pinned memory: the host spins instead of sleeping in hipStreamSynchronize
                if (!c.h_done.p) { c.h_done.need(16); c.h_done.p[0] = 0; }
                c.done_seq = c.done_seq >= 0x7ffffff0 ? 1 : c.done_seq + 1;
                pro.done = c.h_done.p; pro.seq = c.done_seq; c.one_wait = c.done_seq;
            }
            gp.h_out.need((size_t)3 * N);
            const UpdArgs upd1{ c.d_perm.p, gp.d_lp.p, nullptr, gp.d_dx.p, gp.h_out.p, gp.h_dx.p, c.h_bad.p };
            if (one_nt >= 1024) hipLaunchKernelGGL(k_inc_one<1024>, dim3(1), dim3(1024), one_lds, s, pro, fl, c.inl, c.dp, tstep, c.d_tab.p + iu_off, iu_n, c.d_tab.p + id_off, id_n, c.d_pool.p, iu_full, g_opt.block_factor, c.d_x.p, upd1, uctx);
            else if (one_nt >= 512) hipLaunchKernelGGL(k_inc_one<512>, dim3(1), dim3(512), one_lds, s, pro, fl, c.inl, c.dp, tstep, c.d_tab.p + iu_off, iu_n, c.d_tab.p + id_off, id_n, c.d_pool.p, iu_full, g_opt.block_factor, c.d_x.p, upd1, uctx);
            else hipLaunchKernelGGL(k_inc_one<256>, dim3(1), dim3(256), one_lds, s, pro, fl, c.inl, c.dp, tstep, c.d_tab.p + iu_off, iu_n, c.d_tab.p + id_off, id_n, c.d_pool.p, iu_full, g_opt.block_factor, c.d_x.p, upd1, uctx);
        } else
            hipLaunchKernelGGL(k_inc_prologue, dim3(1), dim3(1024), 0, s, pro, fl, c.inl);
    }
    if (iu && !one) {
        const int *list = c.d_tab.p + iu_off;
        if (iu_nt >= 1024) hipLaunchKernelGGL(k_front_small<1024>, dim3(iu_n), dim3(1024), iu_lds, s, c.dp, list, c.d_pool.p, c.d_H.p, c.d_bad.p, iu_full, g_opt.block_factor, c.d_flags.p, 1, uctx);
        else if (iu_nt >= 512) hipLaunchKernelGGL(k_front_small<512>, dim3(iu_n), dim3(512), iu_lds, s, c.dp, list, c.d_pool.p, c.d_H.p, c.d_bad.p, iu_full, g_opt.block_factor, c.d_flags.p, 1, uctx);
        else hipLaunchKernelGGL(k_front_small<256>, dim3(iu_n), dim3(256), iu_lds, s, c.dp, list, c.d_pool.p, c.d_H.p, c.d_bad.p, iu_full, g_opt.block_factor, c.d_flags.p, 1, uctx);
    }
    for (int l = 0; l < nLev; l++) {
        if (lev_dirty[l].empty() || iu || one) continue;
        if (mp && l >= 1) {
            if (l > 1) continue;
            const int *list = c.d_tab.p + mp_up_off;
            if (mp_nt >= 1024) hipLaunchKernelGGL(k_front_small<1024>, dim3(mp_n), dim3(1024), mp_up_lds, s, c.dp, list, c.d_pool.p, c.d_H.p, c.d_bad.p, mp_full, g_opt.block_factor, c.d_flags.p, 1);
            else if (mp_nt >= 512) hipLaunchKernelGGL(k_front_small<512>, dim3(mp_n), dim3(512), mp_up_lds, s, c.dp, list, c.d_pool.p, c.d_H.p, c.d_bad.p, mp_full, g_opt.block_factor, c.d_flags.p, 1);
            else hipLaunchKernelGGL(k_front_small<256>, dim3(mp_n), dim3(256), mp_up_lds, s, c.dp, list, c.d_pool.p, c.d_H.p, c.d_bad.p, mp_full, g_opt.block_factor, c.d_flags.p, 1);
            continue;
        }
        const LevelPlan &L = dl[l];
        if (L.n_small) launch_front_small(c, L, s);
        if (L.n_big) {
            if (g_opt.tile_assembly)
                hipLaunchKernelGGL(k_assemble_tile, dim3(L.asm_tile.grid), dim3(TPB), at_lds(), s, c.dp, c.d_tab.p + L.asm_tile.list_off,
                                   c.d_tab.p + L.asm_tile.pre_off, L.asm_tile.n, c.d_pool.p, c.d_H.p);
            else
                hipLaunchKernelGGL(k_assemble_big, dim3(L.asm_big.grid), dim3(TPB), L.asm_lds, s, c.dp, c.d_tab.p + L.asm_big.list_off,
                                   c.d_tab.p + L.asm_big.pre_off, L.asm_big.n, c.d_pool.p, c.d_H.p);
            enqueue_big_steps(c, L, s, [](int) {}, []() {});
        }
    }
    if (mp && g_opt.wave_backsolve && mp_dn_maxns <= BSW_MAX_NS) hipLaunchKernelGGL(k_backsolve_w, dim3(mp_n), dim3(TPB), mp_dn_lds, s, c.dp, c.d_tab.p + mp_dn_off, c.d_pool.p, c.d_x.p, c.d_flags.p + nFr, c.d_bad.p, UpdArgs{});
    else if (mp) hipLaunchKernelGGL((k_backsolve_t<true>), dim3(mp_n), dim3(TPB), mp_dn_lds, s, c.dp, c.d_tab.p + mp_dn_off, c.d_pool.p, c.d_x.p, 0, c.d_flags.p + nFr, 1, c.d_bad.p, UpdArgs{});
    // incremental steps: the state update (state = l_point + dx, pinned mirrors of state / dx / failure record) rides on the
    // back substitution of the front that owns the pose -- every visited pose lives in a front of this sweep -- instead of
    // a launch of its own over all poses
    // (the device states are NOT touched: d_state / d_lp keep mirroring the host objects, the new states go to a pinned buffer
    // of their own -- see pack_states_diff)
    gp.h_out.need((size_t)3 * N);
    const UpdArgs upd = batch ? UpdArgs{} : UpdArgs{ c.d_perm.p, gp.d_lp.p, nullptr, gp.d_dx.p, gp.h_out.p, gp.h_dx.p, c.h_bad.p };
    bool rode = one;
    if (id && !one) {
        if (g_opt.wave_backsolve && id_maxns <= BSW_MAX_NS) hipLaunchKernelGGL(k_backsolve_w, dim3(id_n), dim3(TPB), id_lds, s, c.dp, c.d_tab.p + id_off, c.d_pool.p, c.d_x.p, c.d_flags.p + nFr, c.d_bad.p, upd);
        else hipLaunchKernelGGL((k_backsolve_t<true>), dim3(id_n), dim3(TPB), id_lds, s, c.dp, c.d_tab.p + id_off, c.d_pool.p, c.d_x.p, 0, c.d_flags.p + nFr, 1, c.d_bad.p, upd);
        rode = true;
    }
    for (int l = (one ? -1 : id ? (needed ? -1 : id_rest) : nLev - 1); l >= 0; l--) {
        if (mp && l >= 1) continue;
        if (batch) { launch_backsolve(c, dl[l], s, [](int) {}, []() {}); continue; }
        if (l >= I.nLev0 || needed) {
            if (bs_n[l] > 0) {
                const size_t lds = l >= I.nLev0 ? solve_lds_of(nF0 + l - I.nLev0) : I.base_levels[l].solve_lds;
                if (g_opt.wave_backsolve && bs_maxns[l] <= BSW_MAX_NS && bs_wlds[l] <= 160 * 1024)      // a few fronts per level: latency is all that counts
                    hipLaunchKernelGGL(k_backsolve_w, dim3((unsigned)bs_n[l]), dim3(TPB), bs_wlds[l], s, c.dp, c.d_tab.p + bs_off[l], c.d_pool.p, c.d_x.p, (int *)nullptr, c.d_bad.p, upd);
                else
                    hipLaunchKernelGGL((k_backsolve_t<false>), dim3((unsigned)bs_n[l]), dim3(TPB), lds, s, c.dp, c.d_tab.p + bs_off[l], c.d_pool.p, c.d_x.p, 0, (int *)nullptr, 0, c.d_bad.p, upd);
                rode = true;
            }
        } else {                                     // every pose is visited: all base fronts, level by level
            const LevelPlan &L = I.base_levels[l];
            hipLaunchKernelGGL((k_backsolve_t<false>), dim3(L.n_all), dim3(TPB), L.solve_lds, s, c.dp, c.d_tab.p + L.all_off, c.d_pool.p, c.d_x.p, 0, (int *)nullptr, 0, c.d_bad.p, upd);
            rode = true;
        }
    }
    if (batch || !rode) {
        hipLaunchKernelGGL(k_update_states, dim3((N + TPB - 1) / TPB), dim3(TPB), 0, s, N, c.d_pos.p, c.d_x.p, gp.d_lp.p, gp.d_state.p, gp.d_dx.p,
                           batch ? gp.h_lp.p : gp.h_out.p, gp.h_dx.p, c.d_bad.p, c.h_bad.p);          // new states / dx / pivot flag straight into pinned memory
        gp.mirror_sync = false;                       // (this kernel rewrites d_state)
    } else {
        gp.mirror_sync = true;                        // device arrays == mirrors == what the host objects held at this call
    }
    gp.new_states = gp.h_out.p;
    HIPCHECK(hipGetLastError());
    if (!batch) { const double te = now_ms(); g_incsub[0] += tsub1 - tsub0; g_incsub[1] += tsub2 - tsub1; g_incsub[2] += tsub3 - tsub2; g_incsub[3] += tsub4 - tsub3; g_incsub[4] += te - tsub4; g_incsub[5] += (double)fd_dirty.size(); g_incsub_n++; }
    if (nT > 0 && I.dirty[nFr - 1]) I.tail_ok = g_opt.inc_tail ? nFr - 1 : -1;      // (re)generated or refactorised by this step, in the padded layout
    for (int t : fd_dirty) I.dirty[t] = 0;
    // the pattern folded into the device structures (a later batch call compares against it)
    c.pat.resize((size_t)2 * F);
    for (int f = Fold; f < F; f++) { c.pat[2 * f] = fa[f]; c.pat[2 * f + 1] = fb[f]; }
    c.patN = N;
    c.st.n_fronts = nFr; c.st.n_levels = nLev;
    return true;
}

// After a synchronised numeric phase: c.h_bad mirrors the device's failure record {flag, front, kind, step}.  kind 9 = a
// dependency-flag poll of a multi-level launch gave up (wait_flag): that is a failure of the launch, not of the matrix, and
// is reported as ERR_DEP_TIMEOUT; everything else is a non-positive pivot (returns true, stats.not_spd).
static bool check_bad(Context &c) {
    if (!c.h_bad.p[0]) { c.st.not_spd = 0; return false; }
    if (c.h_bad.p[0] == 9 || c.h_bad.p[2] == 9) fail(ERR_DEP_TIMEOUT, "a multi-level launch gave up waiting for a dependency flag (the fronts it waits for never finished)");
    c.st.not_spd = 1;
    return true;
}

static void set_lambda(Context &c, GraphPack &gp, double lambda) {
    const int N = c.plan.N;
    if (c.lambda_N == N && c.lambda_val == lambda) return;       // d_lambda already holds it (warm calls)
    c.lambda_N = N; c.lambda_val = lambda;
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
    // Warm call on an unchanged graph (same factor and node counts as packed, plan and device copies current): the pass over
    // the factor objects that finds what the caller edited in place -- reference semantics: every z / W is read on every call;
    // 26 us of pointer chasing on M3500 -- runs WHILE the GPU works on the step, launched on the packed copies.  If the pass
    // finds an edit (or a different factor behind a pointer), the speculative step is thrown away and the call starts over
    // from the fresh copies: nothing of the first run is visible (its inputs are the pinned state mirror, which it does not
    // write; its outputs are overwritten).
    const bool timing0 = g_opt.device_timing != 0;
    bool speculate = g_opt.speculate_factors && !g_opt.trust_factor_cache && !timing0 && c.have_plan && gp.F > 0 && gp.Fg == zsize(g->factors) && gp.N == zsize(g->nodes) &&
                     (int)gp.fptr.size() == gp.Fg && gp.host_idx.empty() && gp.F_on_device == gp.F && gp.dirty_hi <= gp.dirty_lo &&
                     c.patN == gp.N && (int)c.pat.size() == 2 * gp.F && c.inc.t_first.empty() && c.plan_persist == launch_table_key() &&
                     c.plan.leaf_nodes == g_opt.leaf_nodes && c.plan_pin == g_opt.pin_last && g_opt.use_graph && !param->show_timing && !c.no_speculation;
    c.no_speculation = false; c.st.reserved1 = 0; c.st.inc_replanned = 0; c.st.inc_old_old_cross = 0;
    if (!speculate) pack_factors(gp, g);
    pack_states(gp, g, false, false);
    gp.mirror_sync = false;                           // (a batch step leaves new states in d_state and in the l_point mirror)
    const int N = gp.N, F = gp.F;
    c.h_bad.need(4); gp.h_dx.need((size_t)3 * N);
    if (!gp.host_idx.empty()) {       // foreign factor types: their eval() reads the host objects, which the reference
        april_graph_node_t **hn = (april_graph_node_t **)g->nodes->data;     // re-linearises first (aprilsam.c:131-135)
        for (int i = 0; i < N; i++) memcpy(hn[i]->l_point, hn[i]->state, 24);
        eval_host_factors(gp, g, 0);
        upload_host_index(gp);
    }
    const double t1 = now_ms();
    const bool timing = g_opt.device_timing != 0;
    // A graph that only GREW since the plan was made (the reference's demo in --batch_update_only mode, examples/
    // aprilsam_demo.c:224-228; the batch fall-backs of an incremental run): instead of a new nested dissection + symbolic
    // analysis + plan upload per call, the appended poses become tail fronts of the existing plan (the machinery of the
    // incremental path) and EVERY front is re-factorised -- batch semantics on an extended plan.  A full re-plan follows
    // when the tail has grown past extend_tail_fronts fronts, or when the topology stops changing (second call in a row).
    bool hybrid = false, reused = false;
    {
        const int patF = (int)c.pat.size() / 2;
        bool ext = g_opt.batch_extend && !timing && c.have_plan && c.inc.ready && gp.host_idx.empty() && N >= c.patN && F >= patF &&
                   c.inc_N == c.patN && c.inc_F == patF && c.plan.leaf_nodes == g_opt.leaf_nodes && c.plan_pin == g_opt.pin_last &&
                   c.plan_persist == launch_table_key();
        for (int i = 0; i < patF && ext; i++) ext = c.pat[2 * i] == gp.h_fa.p[i] && c.pat[2 * i + 1] == gp.h_fb.p[i];
        const bool grew = ext && (N > c.patN || F > patF);
        if (grew) { c.want_inc = true; c.same_topo_batches = 0; }        // (plans made from now on reserve the append slack)
        else if (ext && !c.inc.t_first.empty()) c.same_topo_batches++;
        const int tails_after = (N - c.inc.Nb + 23) / 24;               // (extend_tail_fronts counts tail fronts of 24 poses, whatever tail_poses is)
        const double lam = param->tikhanov > 0 ? param->tikhanov : 0.0;
        if (ext && N > c.inc.Nb && c.inc.cap_nodes > 0 && tails_after <= g_opt.extend_tail_fronts && (grew || (!c.inc.t_first.empty() && c.same_topo_batches <= 1))) {
            // z / W of already-packed factors edited in place by the caller (pack_factors recorded the range) only reach the
            // device through upload_factors: the patch list of inc_fast_step carries the NEW factors alone
            if (F > gp.F_cap || gp.dirty_hi > gp.dirty_lo) upload_factors(gp);
            c.h_bad.need(4);
            hybrid = inc_fast_step(c, gp, N, F, c.inc_F, c.inc_N, nullptr, lam);
            reused = hybrid;
        }
    }
    double t2 = now_ms(), t3 = t2;
    if (!hybrid && speculate) {
        speculate = prepare_plan(c, gp, g);             // (true: the cached plan fits the packed pattern -- it does, by the conditions above)
        if (speculate) {
            set_lambda(c, gp, param->tikhanov);
            t2 = t3 = now_ms();                         // (stats: the pass over the factor objects below counts as device time -- it runs under it)
            c.h_bad.p[0] = c.h_bad.p[1] = c.h_bad.p[2] = c.h_bad.p[3] = 0;
            run_numeric(c, gp, false, false, true);
            const long long v0 = gp.content_version; const int dev0 = gp.F_on_device;
            pack_factors(gp, g);                        // ... the pass over the factor objects, under the GPU's work
            if (gp.content_version != v0 || gp.F_on_device != dev0 || gp.dirty_hi > gp.dirty_lo || !gp.host_idx.empty()) {
                HIPCHECK(hipStreamSynchronize(gp.stream));      // an edit: this run is void, the call starts over on the fresh copies
                c.no_speculation = true;
                batch_impl(g, param);
                c.st.reserved1 = 1;                             // (stats: this call ran twice)
                return;
            }
            reused = true;
        } else pack_factors(gp, g);
    }
    if (!hybrid && !speculate) {
        reused = prepare_plan(c, gp, g);
        t2 = now_ms();
        upload_factors(gp);
        set_lambda(c, gp, param->tikhanov);
        t3 = now_ms();
        // One graph launch: k_load_states pulls the packed states from the pinned mirror (state and, every node being
        // re-linearised first, aprilsam.c:131-135, l_point), ..., k_update_states leaves new states (h_lp), dx and the pivot
        // flag in pinned mirrors.  No copy-engine call on the path.
        c.h_bad.p[0] = c.h_bad.p[1] = c.h_bad.p[2] = c.h_bad.p[3] = 0;          // (the kernels only ever write a SET failure record)
        run_numeric(c, gp, timing, false, true);
    }
    // while the GPU works: a param that is used incrementally needs the reference's elimination tree of THIS batch step for
    // its next april_graph_cholesky_inc (refmodel.cpp: the reference's own min-degree order, ~1 ms of integer work on M3500)
    bool model_ready = false;
    if (c.used_inc && gp.host_idx.empty()) { c.model.batch(N, F, gp.h_fa.p, gp.h_fb.p); model_ready = true; }      // (not for params that only ever see batch calls)
    // ... and the part of the write-back that does not wait for the result: every node is re-linearised at the state it came
    // in with before anything is solved (aprilsam.c:131-135: l_point = state, whatever the factorisation says later), UID = index
    // (aprilsam.c:628).  The walk also pulls the node objects into the cache for the second half below.
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    for (int i = N - 1; i >= 0; i--) { april_graph_node_t *n = ns[i]; n->UID = i; memcpy(n->l_point, gp.h_state.p + (size_t)3 * i, 24); }
    HIPCHECK(hipStreamSynchronize(gp.stream));
    const double t4 = now_ms();
    check_bad(c);
    c.st.error_code = 0;
    if (c.st.not_spd) {
        c.model.valid = false;
        static bool warned = false;
        if (!warned) { fprintf(stderr, "aprilsam_amd: information matrix not positive definite; node states left untouched\n"); warned = true; }
    } else {
        // write back: state / delta_X where not NaN-skipped (l_point and UID went in above)
        for (int i = N - 1; i >= 0; i--) {                                   // aprilsam.c:311-315 order
            april_graph_node_t *n = ns[i];
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
        param->factor_num = gp.Fg;      // (graph factors; F counts packed entries, pack_factors)
        c.have_fact = true; c.batch_nodes = N; c.batch_factors = F; c.model.valid = model_ready;
        if (!hybrid) inc_prepare(c);                     // (an extended plan keeps its base + tail bookkeeping)
        c.inc_F = F; c.inc_N = N;
        record_unary_points(gp, 0, F, gp.h_state.p);         // the linearisation point of this call
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
        fflush(stdout);
    }
}

void batch_step(april_graph_t *g, april_graph_cholesky_param_t *param) {
    if (zsize(g->nodes) == 0 || zsize(g->factors) == 0) return;          // aprilsam.c:90-91
    guarded(param, g, [&] {
        if (!param->nreordering) fail(ERR_UNSUPPORTED, "april_graph_cholesky: param->nreordering == 0 (the reference asserts, aprilsam.c:372-374)");
        ensure_device();
        std::lock_guard<std::mutex> lk(g_mu);
        batch_impl(g, param);
    });
}

// ------------------------------------------------------------------------------------------------------
// incremental step (aprilsam.c:377-576).  The linear system the reference maintains by partial un-/re-
// factorisation — every factor linearised at its nodes' l_point (aprilsam.c:508-542; l_points only move in
// a batch step), Tikhonov term only on poses present at the last batch step (aprilsam.c:197-204 vs :508-542)
// — is solved on the GPU (only the fronts on the root paths of the new factors are re-assembled and re-factorised,
// inc_fast_step); WHICH poses receive the result,
// the relinearisation counter and the batch fall-back follow the reference exactly through the bookkeeping
// model of refmodel.cpp (measured: on the poses it touches, the reference's result is the exact solution).
// ------------------------------------------------------------------------------------------------------
// APRILSAM_AMD_INC_PROFILE=1: host wall-clock split of the incremental steps, printed at process exit
struct IncProf {
    bool on = false; double acc[8] = { 0 }; long long n = 0;
    std::vector<std::array<float, 7>> steps;          // per step: the six phases + total (medians at exit)
    std::vector<std::array<float, 4>> kst;            // =2: phases of k_inc_one in us (patches, linearise, fronts, back substitution)
    std::vector<std::array<float, 10>> fst;           //     ... and of its last front
    IncProf() { const char *e = getenv("APRILSAM_AMD_INC_PROFILE"); on = e && (*e == '1' || *e == '2'); }
    ~IncProf() {
        if (!on || !n) return;
        fprintf(stderr, "aprilsam_amd inc profile over %lld steps (ms/step): pack %.4f model %.4f upload %.4f plan+enqueue %.4f d2h+sync %.4f writeback %.4f | total %.4f\n",
                n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, (acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5]) / n);
        double med[7];
        for (int k = 0; k < 7; k++) {
            std::vector<float> v(steps.size());
            for (size_t i = 0; i < steps.size(); i++) v[i] = steps[i][k];
            std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
            med[k] = v[v.size() / 2];
        }
        fprintf(stderr, "aprilsam_amd inc profile, MEDIANS (ms): pack %.4f model %.4f upload %.4f plan+enqueue %.4f d2h+sync %.4f writeback %.4f | total %.4f\n",
                med[0], med[1], med[2], med[3], med[4], med[5], med[6]);
        fprintf(stderr, "aprilsam_amd inc profile, states: %lld steps; all states loaded because the mirrors were not in step %lld, the library's own updates > 48 poses %lld, the caller's changes > 48 poses %lld\n",
                g_full_reason[3], g_full_reason[0], g_full_reason[1], g_full_reason[2]);
        if (g_incsub_n) fprintf(stderr, "aprilsam_amd inc profile, general path over %lld steps (us/step): owners %.2f regenerate fronts %.2f launch tables %.2f patches %.2f enqueue %.2f | fronts regenerated per step %.1f | steps eligible for tail_refactor %.0f, of them with a short walk and patched states %.0f\n",
                                g_incsub_n, 1e3 * g_incsub[0] / g_incsub_n, 1e3 * g_incsub[1] / g_incsub_n, 1e3 * g_incsub[2] / g_incsub_n, 1e3 * g_incsub[3] / g_incsub_n, 1e3 * g_incsub[4] / g_incsub_n, g_incsub[5] / g_incsub_n, g_incsub[6], g_incsub[7]);
        fprintf(stderr, "aprilsam_amd inc profile, low-rank updates: %lld general-path steps with updated fronts (%lld fronts updated, %lld re-factorised), %lld without (%lld fronts re-factorised); %lld of all of them as one launch\n",
                g_updstat[0], g_updstat[2], g_updstat[3], g_updstat[1], g_updstat[4], g_updstat[5]);
        { std::string r; for (int k = 0; k < 32; k++) if (g_incfail[k]) r += " #" + std::to_string(k) + ":" + std::to_string(g_incfail[k]);
          fprintf(stderr, "aprilsam_amd inc profile, steps handed to a full re-plan by exit of inc_fast_step:%s\n", r.empty() ? " none" : r.c_str()); }
        if (!kst.empty()) {
            double km[4];
            for (int k = 0; k < 4; k++) {
                std::vector<float> v(kst.size());
                for (size_t i = 0; i < kst.size(); i++) v[i] = kst[i][k];
                std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
                km[k] = v[v.size() / 2];
            }
            fprintf(stderr, "aprilsam_amd inc profile, k_inc_one over %zu steps, MEDIANS (us): patches %.2f linearise %.2f fronts %.2f back substitution + update %.2f\n",
                    kst.size(), km[0], km[1], km[2], km[3]);
            double fm[10];
            for (int k = 0; k < 10; k++) {
                std::vector<float> v(fst.size());
                for (size_t i = 0; i < fst.size(); i++) v[i] = fst[i][k];
                std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
                fm[k] = v[v.size() / 2];
            }
            fprintf(stderr, "aprilsam_amd inc profile, last front of k_inc_one, MEDIANS (us): zero %.2f records %.2f work lists %.2f extend-add %.2f factorise %.2f store %.2f | "
                    "own poses %.0f struct poses %.0f children %.0f work-list entries %.0f\n", fm[0], fm[1], fm[2], fm[3], fm[4], fm[5], fm[6], fm[7], fm[8], fm[9]);
        }
    }
};
static IncProf g_incprof;

static void inc_impl(april_graph_t *g, april_graph_cholesky_param_t *param);
static void apply_visits(Context &c, GraphPack &gp, april_graph_t *g, april_graph_cholesky_param_t *param, int N);
void inc_step(april_graph_t *g, april_graph_cholesky_param_t *param) {
    if (zsize(g->nodes) == 0 || zsize(g->factors) == 0) return;          // aprilsam.c:380-381
    guarded(param, g, [&] { inc_impl(g, param); });
}
static void inc_impl(april_graph_t *g, april_graph_cholesky_param_t *param) {
    std::lock_guard<std::mutex> lk(g_mu);
    {
        auto it = g_ctx.find(param);
        if (it == g_ctx.end() || !it->second->have_fact) return;         // aprilsam.c:382-383 (no prior chol)
    }
    if (param->factor_num == zsize(g->factors)) return;                  // aprilsam.c:384-385
    ensure_device();
    Context &c = ctx_for(param);
    GraphPack &gp = pack_for(g);
    c.want_inc = true; c.used_inc = true;
    const double t0 = now_ms();
    pack_factors(gp, g, false);
    const int N = zsize(g->nodes), F = gp.F;
    c.h_bad.need(4);
    const double tp0a = now_ms();
    if (!c.model.valid) c.model.batch(c.batch_nodes, c.batch_factors, gp.h_fa.p, gp.h_fb.p);      // lazily, after a batch step
    c.model.inc_begin(N, F, gp.h_fa.p, gp.h_fb.p);
    std::vector<RefModel::Visit> &visits = c.visits;
    c.model.plan_visit(visits);                      // structural: which poses the reference's solve_node touches
    const bool partial = c.model.naffected <= 5;     // aprilsam.c:755: otherwise the whole tree is walked
    const double tp0b = now_ms();
    // states: the pinned mirrors follow the node objects; the fast path patches / loads the device copies from its first kernels.
    // A partial walk reads only the poses of the new factors and the visited ones: only those are looked at (pack_states_some)
    bool lazy_states = false, patch_states;
    if (partial && g_opt.inc_lazy_states) {
        std::vector<int> &inv = c.involved; inv.clear();
        for (int f = c.inc_F; f < F; f++) { inv.push_back(gp.h_fa.p[f]); if (gp.h_fb.p[f] >= 0) inv.push_back(gp.h_fb.p[f]); }
        for (const RefModel::Visit &v : visits) inv.push_back(v.node);
        patch_states = pack_states_some(gp, g, inv); lazy_states = true;
    } else patch_states = pack_states_diff(gp, g);
    const double tp1 = now_ms() - (tp0b - tp0a);      // (profile: "pack" = factors + states, "model" = the bookkeeping in between)
    const double tp2 = tp1 + (tp0b - tp0a);
    if (F > gp.F_cap || !g_opt.inc_fast || !gp.host_idx.empty()) upload_factors(gp);     // (growing the device arrays re-uploads everything)
    if (!gp.host_idx.empty()) {       // new foreign factors are linearised now, at the host objects' current l_points
        eval_host_factors(gp, g, gp.host_evaluated);     // (aprilsam.c:508-542); older ones keep their evaluation
        upload_host_index(gp);
    }
    const double tp3 = now_ms();
    record_unary_points(gp, c.inc_F, F, gp.h_state.p);  // priors added by this call are evaluated at their node's state now
    // fast path: frozen base plan + TAIL front, only the dirty root paths are regenerated and re-factorised
    const int N_before = c.inc_N;
    c.h_bad.p[0] = c.h_bad.p[1] = c.h_bad.p[2] = c.h_bad.p[3] = 0;      // (the riding state update only ever writes a SET failure record)
    bool reused = g_opt.inc_fast && gp.host_idx.empty() && inc_fast_step(c, gp, N, F, c.inc_F, c.inc_N, partial ? &visits : nullptr, -1.0, patch_states);
    if (!reused) {                // the step does not fit the frozen structure (or slack ran out): full re-plan
        if (lazy_states) pack_states(gp, g, true, false);       // (every pose's state / l_point goes to the device below: look at all of them)
        gp.mirror_sync = false; gp.new_states = gp.h_state.p;
        upload_factors(gp);
        HIPCHECK(hipMemcpyAsync(gp.d_state.p, gp.h_state.p, (size_t)24 * N, hipMemcpyHostToDevice, gp.stream));
        HIPCHECK(hipMemcpyAsync(gp.d_lp.p, gp.h_lp.p, (size_t)24 * N, hipMemcpyHostToDevice, gp.stream));
        gp.d_upt.need((size_t)3 * F);
        HIPCHECK(hipMemcpyAsync(gp.d_upt.p, gp.h_upt.data(), (size_t)24 * F, hipMemcpyHostToDevice, gp.stream));
        prepare_plan(c, gp, g);
        c.h_lambda.assign(N, 0.0);
        for (int i = 0; i < N; i++) if (c.plan.perm[i] < c.batch_nodes && param->tikhanov > 0) c.h_lambda[i] = param->tikhanov;
        HIPCHECK(hipMemcpyAsync(c.d_lambda.p, c.h_lambda.data(), (size_t)8 * N, hipMemcpyHostToDevice, gp.stream));
        c.lambda_N = -1;                                  // (not the uniform batch value)
        run_numeric(c, gp, false, true);
        inc_prepare(c);
    }
    c.inc_F = F; c.inc_N = N; c.same_topo_batches = 0;
    if (!reused) c.st.inc_fronts_updated = 0;
    c.st.inc_replanned = reused ? 0 : 1; c.st.inc_old_old_cross = c.model.old_old_cross;      // (include/aprilsam_amd.h: what the caller is told)
    const double tp4 = now_ms();
    if (!reused) {                // (the fast path's last kernel wrote states, dx and the pivot flag into the pinned mirrors itself)
        HIPCHECK(hipMemcpyAsync(gp.h_state.p, gp.d_state.p, (size_t)24 * N, hipMemcpyDeviceToHost, gp.stream));
        HIPCHECK(hipMemcpyAsync(gp.h_dx.p, gp.d_dx.p, (size_t)24 * N, hipMemcpyDeviceToHost, gp.stream));
        HIPCHECK(hipMemcpyAsync(c.h_bad.p, c.d_bad.p, 16, hipMemcpyDeviceToHost, gp.stream));
    }
    bool arrived = false;
    if (reused && c.one_wait) {          // k_inc_one wrote everything else before this word; a launch that never answers is left to hipStreamSynchronize
        const volatile int *w = c.h_done.p;
        const double tw0 = now_ms();
        for (int spins = 0; !(arrived = (*w == c.one_wait)); spins++) if ((spins & 1023) == 1023 && now_ms() - tw0 > 2.0) break;
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!arrived) HIPCHECK(hipStreamSynchronize(gp.stream));
    c.one_wait = 0;
    const double tp5 = now_ms();
    if (g_incprof.on && g_incprof_stamps && reused && c.h_kstamp.p && c.h_kstamp.p[4]) {
        const long long *k = c.h_kstamp.p;
        g_incprof.kst.push_back({ (float)(k[1] - k[0]) * 0.01f, (float)(k[2] - k[1]) * 0.01f, (float)(k[3] - k[2]) * 0.01f, (float)(k[4] - k[3]) * 0.01f });
        const long long *f = k + 8;                  // last front of the step: zero, records, work lists, extend-add, factorisation, store; dims
        g_incprof.fst.push_back({ (float)(f[4] - f[0]) * 0.01f, (float)(f[5] - f[4]) * 0.01f, f[6] ? (float)(f[6] - f[5]) * 0.01f : 0.f, (float)(f[1] - (f[6] ? f[6] : f[5])) * 0.01f,
                                  (float)(f[2] - f[1]) * 0.01f, (float)(f[3] - f[2]) * 0.01f, (float)k[5], (float)k[6], (float)k[7], (float)f[7] });
        c.h_kstamp.p[4] = 0;
    }
    check_bad(c);
    c.st.error_code = 0;
    c.st.n_nodes = N; c.st.n_factors = F; c.st.symbolic_reused = reused;
    if (c.st.not_spd) {
        static bool warned = false;
        if (!warned) { fprintf(stderr, "aprilsam_amd: incremental system not positive definite; node states left untouched\n"); warned = true; }
        c.inc.tail_ok = -1;                          // (a front stopped half-way: nothing to refactorise from)
        c.inc.upd_ok = false;                        // (... nor to update)
        return;
    }
    // bookkeeping exactly as the reference: which poses solve_node visits / updates, start_over (refmodel.cpp)
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    for (int i = (partial && reused) ? std::min(N_before, N) : 0; i < N; i++) ns[i]->UID = i;      // aprilsam.c:474 (the new nodes; every node where the walk is full anyway)
    const int start_over_before = c.model.start_over;
    apply_visits(c, gp, g, param, N);
    if (param->ordering) free(param->ordering);
    param->ordering = (int *)malloc(sizeof(int) * (size_t)N);
    memcpy(param->ordering, c.plan.perm.data(), sizeof(int) * (size_t)N);
    param->nreordering = N;
    param->factor_num = gp.Fg;
    const double step_ms = now_ms() - t0;
    c.st.ms_total = step_ms;
    if (param->show_timing)
        printf("aprilsam_amd inc: N=%d F=%d fronts=%d (%d regenerated) marked=%d visited=%zu%s | pack %.3f model %.3f plan+enqueue %.3f device %.3f | total %.3f ms\n",
               N, F, c.st.n_fronts, reused ? c.st.reserved0 : c.st.n_fronts, c.model.naffected, c.visits.size(), reused ? "" : " (re-planned)",
               tp1 - t0, tp2 - tp1, tp4 - tp3, tp5 - tp4, step_ms), fflush(stdout);
    if (g_incprof.on) {
        g_incprof.acc[0] += tp1 - t0; g_incprof.acc[1] += tp2 - tp1; g_incprof.acc[2] += tp3 - tp2; g_incprof.acc[3] += tp4 - tp3;
        const double te = now_ms();
        g_incprof.acc[4] += tp5 - tp4; g_incprof.acc[5] += te - tp5; g_incprof.n++;
        g_incprof.steps.push_back({ (float)(tp1 - t0), (float)(tp2 - tp1), (float)(tp3 - tp2), (float)(tp4 - tp3), (float)(tp5 - tp4), (float)(te - tp5), (float)(te - t0) });
    }
    // aprilsam.c:557-559, the wall-clock rule: "this step took longer than a third of a batch step -> start over".  The
    // reference sets start_over = INT_MAX BEFORE its solver call, whose walk then adds one per pose that newly crossed the
    // relinearisation threshold (:741-747): with at least one such pose the counter wraps negative and the fall-back does
    // NOT happen (nor any threshold fall-back until the rule fires again).  Reproduced as is: same inputs, same schedule.
    if (!g_opt.deterministic && step_ms > param->batch_time / 3)
        c.model.start_over = (int)(0x7fffffffu + (unsigned)(c.model.start_over - start_over_before));
    if (c.model.start_over > param->nthreshold) {                                                   // aprilsam.c:566-575
        const double b0 = now_ms();
        const int rp = c.st.inc_replanned, oc = c.st.inc_old_old_cross;
        batch_impl(g, param);
        c.st.inc_replanned = rp; c.st.inc_old_old_cross = oc;       // (they describe the incremental step this call made first)
        param->batch_time = now_ms() - b0;
    }
}

// Back substitution over the CURRENT structures of a param (base plan of the last batch step + tail fronts appended since),
// restricted to the fronts that hold a pose of `needed` and their ancestors (null: every front).  Used by
// april_graph_cholesky_inc_solver; april_graph_cholesky_inc has the same loop inside inc_fast_step, fed by its patch list.
static void enqueue_backsolve_current(Context &c, GraphPack &gp, const std::vector<RefModel::Visit> *needed) {
    IncState &I = c.inc; const Plan &P = c.plan;
    const int nF0 = I.nF0, nT = (int)I.t_first.size(), nFr = nF0 + nT, nLev = I.nLev0 + nT, Nb = I.Nb;
    I.need.assign(nFr, needed ? 0 : 1);
    if (needed)
        for (const RefModel::Visit &v : *needed) {
            int t = v.node >= Nb ? I.tf_of[v.node - Nb] : I.pos_front[P.pos[v.node]];
            while (t >= 0 && !I.need[t]) { I.need[t] = 1; t = I.parent[t]; }
        }
    std::vector<int> &tab = c.solve_tab; tab.clear();
    std::vector<int> off(nLev + 1, 0);
    {
        std::vector<int> cnt(nLev, 0);
        for (int t = 0; t < nFr; t++) if (I.need[t]) cnt[I.f_level[t]]++;
        for (int l = 0; l < nLev; l++) off[l + 1] = off[l] + cnt[l];
        tab.resize(std::max(1, off[nLev]));
        std::vector<int> fill(off.begin(), off.end() - 1);
        for (int t = 0; t < nFr; t++) if (I.need[t]) tab[fill[I.f_level[t]]++] = t;
    }
    c.d_solve_tab.need(tab.size());
    hipStream_t s = gp.stream;
    HIPCHECK(hipMemcpyAsync(c.d_solve_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));      // (c.solve_tab lives until the caller's sync)
    for (int l = nLev - 1; l >= 0; l--) {
        const int n = off[l + 1] - off[l];
        if (!n) continue;
        int maxns = 0; size_t wlds = 0, tlds = 0;
        for (int k = off[l]; k < off[l + 1]; k++) {
            const FrontDesc &D = I.fd[tab[k]];
            const int ns = 3 * D.nsb, m = 3 * (D.nsb + D.nub);
            maxns = std::max(maxns, ns); wlds = std::max(wlds, backsolve_lds(m, ns, true));
            tlds = std::max(tlds, (size_t)(m + NB + 8 + NB * (NB + 1)) * 8);
        }
        if (g_opt.wave_backsolve && maxns <= BSW_MAX_NS && wlds <= 160 * 1024)
            hipLaunchKernelGGL(k_backsolve_w, dim3((unsigned)n), dim3(TPB), wlds, s, c.dp, c.d_solve_tab.p + off[l], c.d_pool.p, c.d_x.p, (int *)nullptr, (int *)nullptr, UpdArgs{});
        else
            hipLaunchKernelGGL((k_backsolve_t<false>), dim3((unsigned)n), dim3(TPB), tlds, s, c.dp, c.d_solve_tab.p + off[l], c.d_pool.p, c.d_x.p, 0, (int *)nullptr, 0, (int *)nullptr, UpdArgs{});
    }
}

// After the numbers arrived (gp.h_dx / gp.h_state hold dx and l_point + dx of every pose the back substitution reached):
// the reference's bookkeeping, aprilsam.c:741-775 -- relinearisation counter over the visited poses, delta_X of every visited
// pose, state of the updated ones (NaN guard april_graph_xyt.c:304-305) -- and param->delta_x, which the reference only
// keeps when the caller pre-allocated it (aprilsam.c:590-595; x is a fresh zero vector per call, :583, so poses the walk did
// not reach read 0; indexed like the unknowns: 3 * position in param->ordering).
static void apply_visits(Context &c, GraphPack &gp, april_graph_t *g, april_graph_cholesky_param_t *param, int N) {
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    const double *x = gp.h_dx.p;                                          // dx per node; NaN where the solve produced NaN
    c.model.count_relinearized(x, param->delta_xy, param->delta_theta, c.visits);
    const size_t nvis = c.visits.size();
    for (size_t vk = 0; vk < nvis; vk++) {
        // (a full walk visits the poses in tree order, i.e. all over the node array: the node object 16 visits ahead and the
        // arrays behind the one 8 ahead are requested now -- three dependent cache misses per pose otherwise)
        if (vk + 16 < nvis) __builtin_prefetch(ns[c.visits[vk + 16].node]);
        if (vk + 8 < nvis) { const april_graph_node_t *n8 = ns[c.visits[vk + 8].node]; __builtin_prefetch(n8->delta_X, 1); __builtin_prefetch(n8->state, 1); }
        const RefModel::Visit &vis = c.visits[vk];
        const int n = vis.node; const bool update = vis.update;
        april_graph_node_t *nd = ns[n];
        const double *dx = x + (size_t)3 * n;
        memcpy(nd->delta_X, dx, 24);                                      // aprilsam.c:752-754
        if (!update) continue;
        if (std::isnan(dx[0]) || std::isnan(dx[1]) || std::isnan(dx[2])) continue;   // april_graph_xyt.c:304-305
        memcpy(nd->state, gp.new_states + (size_t)3 * n, 24);             // l_point + dx, theta wrapped (state update on the device)
        // the pinned mirror follows right here (the next call's walk over the node objects then finds this pose unchanged instead
        // of copying it again -- after a full walk that is every pose); the device copy is brought up to date by that call
        // (a pose whose new state equals the old one bit for bit -- most of a full walk: the far past does not move -- needs nothing)
        if (gp.mirror_sync && gp.new_states != gp.h_state.p && memcmp(gp.h_state.p + (size_t)3 * n, gp.new_states + (size_t)3 * n, 24) != 0) {
            memcpy(gp.h_state.p + (size_t)3 * n, gp.new_states + (size_t)3 * n, 24); gp.pending.push_back(n);
        }
    }
    if (param->delta_x) {
        free(param->delta_x);
        param->delta_x = (double *)calloc((size_t)3 * N, sizeof(double));
        for (const RefModel::Visit &vis : c.visits) memcpy(param->delta_x + (size_t)3 * c.plan.pos[vis.node], x + (size_t)3 * vis.node, 24);
    }
}

// aprilsam.c:578-597: back substitution + state update on the current factorisation, with solve_node's visit rule
// (aprilsam.c:721-779: after the last april_graph_cholesky_inc marked more than 5 poses the whole tree is walked and every
// pose gets state = l_point + x -- the caller's CURRENT l_points, april_graph_xyt.c:307-308; otherwise only the root is
// reached and only its delta_X is written).  y persists inside the fronts (the right-hand-side row), so x is reproducible.
void inc_solve_only(april_graph_t *g, april_graph_cholesky_param_t *param) {
    if (zsize(g->nodes) == 0) return;
    guarded(param, g, [&] {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_ctx.find(param);
        if (it == g_ctx.end() || !it->second->have_fact || !param->nreordering) return;        // aprilsam.c:580
        ensure_device();
        Context &c = *it->second;
        GraphPack &gp = pack_for(g);
        const int N = c.inc_N;
        // poses added since the factorisation was made are april_graph_cholesky_inc's business (the reference would read past
        // the end of its factor here)
        if (!c.inc.ready || zsize(g->nodes) != N || gp.N != N) return;
        const double t0 = now_ms();
        pack_states(gp, g, true, false);
        c.h_bad.need(4);
        if (!c.model.valid) c.model.batch(c.batch_nodes, c.batch_factors, gp.h_fa.p, gp.h_fb.p);
        c.model.plan_visit(c.visits);
        const bool partial = c.model.naffected <= 5;
        hipStream_t s = gp.stream;
        set_small_attr();
        hipLaunchKernelGGL(k_load_states_lp, dim3((3 * N + TPB - 1) / TPB), dim3(TPB), 0, s, 3 * N, gp.h_state.p, gp.h_lp.p, gp.d_state.p, gp.d_lp.p, c.d_bad.p);
        enqueue_backsolve_current(c, gp, partial ? &c.visits : nullptr);
        hipLaunchKernelGGL(k_update_states, dim3((N + TPB - 1) / TPB), dim3(TPB), 0, s, N, c.d_pos.p, c.d_x.p, gp.d_lp.p, gp.d_state.p, gp.d_dx.p,
                           gp.h_state.p, gp.h_dx.p, c.d_bad.p, c.h_bad.p);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(s));
        gp.mirror_sync = false; gp.new_states = gp.h_state.p;           // (k_update_states rewrote d_state and the state mirror)
        apply_visits(c, gp, g, param, N);
        c.st.error_code = 0; c.st.ms_total = now_ms() - t0;
        if (param->show_timing) { printf("aprilsam_amd solve: N=%d visited %zu poses%s | total %.3f ms\n", N, c.visits.size(), partial ? " (marked root paths only)" : "", c.st.ms_total); fflush(stdout); }
    });
}

static double chi2_impl(april_graph_t *g);
double graph_chi2(april_graph_t *g) {
    if (zsize(g->factors) == 0) return 0;
    double out = std::nan("");                        // a failed evaluation (errors.h) returns NaN
    guarded(nullptr, g, [&] { out = chi2_impl(g); });
    return out;
}
static double chi2_impl(april_graph_t *g) {
    ensure_device();
    std::lock_guard<std::mutex> lk(g_mu);
    GraphPack &gp = pack_for(g);
    pack_factors(gp, g);
    pack_states(gp, g, false);
    upload_factors(gp);
    double chi2 = device_chi2(gp);
    if (!gp.host_idx.empty()) {       // april_graph.c:90-93: factors other than xyt contribute eval()->chi2
        april_graph_factor_t **fs = (april_graph_factor_t **)g->factors->data;
        int last = -1;
        for (int idx : gp.host_idx) {
            const int gi = gp.p2g[idx];
            if (gi == last) continue;                      // (the pairs of a factor with more than two nodes: one evaluation)
            last = gi;
            april_graph_factor_eval_t *e = fs[gi]->eval(fs[gi], g, nullptr);
            chi2 += e->chi2;
            april_graph_factor_eval_destroy(e);
        }
    }
    return chi2;
}

// ------------------------------------------------------------------------------------------------------
// device-resident driver API: states never leave HBM between Gauss-Newton steps
// ------------------------------------------------------------------------------------------------------
static int resident_begin_impl(april_graph_t *g, april_graph_cholesky_param_t *param);
int resident_begin(april_graph_t *g, april_graph_cholesky_param_t *param) { return guarded_rc(param, g, [&] { return resident_begin_impl(g, param); }); }
static int resident_begin_impl(april_graph_t *g, april_graph_cholesky_param_t *param) {
    if (zsize(g->nodes) == 0 || zsize(g->factors) == 0) return -1;
    ensure_device();
    std::lock_guard<std::mutex> lk(g_mu);
    Context &c = ctx_for(param);
    GraphPack &gp = pack_for(g);
    pack_factors(gp, g);
    if (!gp.host_idx.empty()) return -4;          // host-evaluated factors need the host in the loop: use april_graph_cholesky
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
static int resident_steps_impl(april_graph_t *g, april_graph_cholesky_param_t *param, int n, int mode);
int resident_steps(april_graph_t *g, april_graph_cholesky_param_t *param, int n, int mode) { return guarded_rc(param, g, [&] { return resident_steps_impl(g, param, n, mode); }); }
static int resident_steps_impl(april_graph_t *g, april_graph_cholesky_param_t *param, int n, int mode) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(param);
    if (it == g_ctx.end() || !it->second->have_plan) return -1;
    Context &c = *it->second;
    GraphPack &gp = pack_for(g);
    hipStream_t s = gp.stream;
    const int N = gp.N;
    HIPCHECK(hipSetDevice(g_device));
    set_small_attr();
    gp.mirror_sync = false;                           // (states move on the device only)
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
static int resident_sync_impl(april_graph_t *g, april_graph_cholesky_param_t *param);
int resident_sync(april_graph_t *g, april_graph_cholesky_param_t *param) { return guarded_rc(param, g, [&] { return resident_sync_impl(g, param); }); }
static int resident_sync_impl(april_graph_t *g, april_graph_cholesky_param_t *param) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(param);
    if (it == g_ctx.end()) return -1;
    Context &c = *it->second;
    GraphPack &gp = pack_for(g);
    HIPCHECK(hipMemcpyAsync(c.h_bad.p, c.d_bad.p, 16, hipMemcpyDeviceToHost, gp.stream));
    HIPCHECK(hipStreamSynchronize(gp.stream));
    check_bad(c);
    if (c.h_bad.p[0] && getenv("APRILSAM_AMD_DEBUG")) {
        const int t = c.h_bad.p[1];
        fprintf(stderr, "aprilsam_amd: bad pivot: front %d kernel %d step %d", t, c.h_bad.p[2], c.h_bad.p[3]);
        if (t >= 0 && t < c.plan.nF) fprintf(stderr, " (nsb %d nub %d level %d off %lld)", c.plan.f_nsb[t], c.plan.f_nub[t], c.plan.f_level[t], (long long)c.plan.f_off[t]);
        fprintf(stderr, "\n");
    }
    return c.h_bad.p[0] ? -2 : 0;
}
static double resident_chi2_impl(april_graph_t *g);
double resident_chi2(april_graph_t *g) {
    double out = std::nan("");
    guarded(nullptr, g, [&] { out = resident_chi2_impl(g); });
    return out;
}
static double resident_chi2_impl(april_graph_t *g) {
    std::lock_guard<std::mutex> lk(g_mu);
    return device_chi2(pack_for(g));
}
static int resident_end_impl(april_graph_t *g, april_graph_cholesky_param_t *param);
int resident_end(april_graph_t *g, april_graph_cholesky_param_t *param) { return guarded_rc(param, g, [&] { return resident_end_impl(g, param); }); }
static int resident_end_impl(april_graph_t *g, april_graph_cholesky_param_t *param) {
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
    param->nreordering = N; param->factor_num = gp.Fg;
    c.have_fact = true; c.batch_nodes = N; c.batch_factors = F; c.model.valid = false;
    inc_prepare(c); c.inc_F = F; c.inc_N = N;
    record_unary_points(gp, 0, F, gp.h_lp.p);                // (unary factors were last linearised at the final l_points)
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
    const size_t small_max = (size_t)g_opt.small_lds_kb * 1024;
    for (int t = 0; t < P.nF; t++) {
        const double ns = 3.0 * P.f_nsb[t], nu = 3.0 * P.f_nub[t], R = P.rows(t), C = P.cols(t);
        double fl = 0;                                          // sum_j c_j^2 over this front's columns (+ rhs row)
        for (int q = 0; q < (int)ns; q++) { double cj = (ns - q) + nu + 1; fl += cj * cj; }
        const int nwp = waves_of(small_threads_for((size_t)(P.lev_ptr[P.f_level[t] + 1] - P.lev_ptr[P.f_level[t]])));
        const bool small = small_front_lds((int)R, (int)C, nwp) <= small_max || (g_opt.panel_mode && panel_front_lds((int)R, (int)ns, nwp) <= small_max);
        // algorithmic bytes of a front: its L panel + update block written once, children's updates read once
        const double by = 8.0 * (ns * (ns + 1) / 2 + (nu + 1) * ns + (nu + 1) * (nu + 2) / 2);
        if (small) { flops[K_FRONT_SMALL] += fl; bytes[K_FRONT_SMALL] += by; }
        else {
            // multi-workgroup path.  k_syrk_big gets exactly what its launches are asked for: per outer block of OBP panels the
            // K = block-width update of the lower trapezoid to the right of the block, 2 K flops per element (the kernel also
            // multiplies the upper halves of its diagonal tiles: executed, not algorithmic, not counted); with right-looking
            // panels (left_panels = 0) also the narrow updates inside the block.  The panel kernel (k_diagpanel_ll) gets the
            // rest of the front's sum c_j^2: diagonal blocks, row solves, the left-looking K <= 96 products.
            const double Rv = R - 2;
            double fsy = 0;
            auto trapezoid = [&](double c_lo, double c_hi) { const double n = c_hi - c_lo; return n <= 0 ? 0.0 : n * Rv - (c_lo + c_hi - 1) * n / 2; };   // elements (i >= j) of columns [c_lo, c_hi), rows < Rv
            const int steps = ((int)ns + NB - 1) / NB;
            for (int o = 0; o * OBP < steps; o++) {
                const double k_lo = (double)o * OBP * NB, k_hi = std::min<double>(ns, (double)(o + 1) * OBP * NB);
                fsy += 2.0 * (k_hi - k_lo) * trapezoid(k_hi, C);
                if (!g_opt.block_panels && !(g_opt.left_panels && g_opt.fused_panel))
                    for (double k1 = k_lo + NB; k1 < k_hi; k1 += NB) fsy += 2.0 * NB * trapezoid(k1, k_hi);
            }
            fsy = std::min(fsy, fl);
            flops[K_SYRK_BIG] += fsy; flops[K_PANEL_BIG] += fl - fsy;
            bytes[K_SYRK_BIG] += by; bytes[K_ASSEMBLE_BIG] += by;      // (assembly: the front written once, children's updates read once)
        }
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

// ------------------------------------------------------------------------------------------------------
// stage-level parity exports (plan items 1-2): what the device linearisation and the gather
// assembly produce, in the caller's node coordinates, for comparison with the reference's own J / r / A / B
// ------------------------------------------------------------------------------------------------------
// what = 0: per factor 33 doubles -- (J_a^T W) J_a (symmetric, full), (J_a^T W) J_b (rows a, columns b), (J_b^T W) J_b,
//           (J_a^T W) r, (J_b^T W) r -- read back from the contribution slots k_linearize wrote (out: 33 * F)
// what = 1: the assembled normal equations A = sum J^T W J + lambda I (dense symmetric (3N)^2, row-major) and
//           B = sum J^T W r (3N) in NODE coordinates, from the per-destination sums of the assembly's own source lists
//           (out: 9 N^2 + 3 N doubles; N <= 2000)
static int debug_stage_impl(april_graph_t *g, april_graph_cholesky_param_t *param, int what, double *out);
int debug_stage(april_graph_t *g, april_graph_cholesky_param_t *param, int what, double *out) { return guarded_rc(param, g, [&] { return debug_stage_impl(g, param, what, out); }); }
static int debug_stage_impl(april_graph_t *g, april_graph_cholesky_param_t *param, int what, double *out) {
    if (zsize(g->nodes) == 0 || zsize(g->factors) == 0) return -1;
    ensure_device();
    std::lock_guard<std::mutex> lk(g_mu);
    Context &c = ctx_for(param);
    GraphPack &gp = pack_for(g);
    pack_factors(gp, g);
    if (!gp.host_idx.empty()) return -4;
    pack_states(gp, g, false);
    prepare_plan(c, gp, g);
    upload_factors(gp);
    const Plan &P = c.plan;
    const int N = gp.N, F = gp.F;
    hipStream_t s = gp.stream;
    gp.mirror_sync = false;
    HIPCHECK(hipMemcpyAsync(gp.d_lp.p, gp.d_state.p, (size_t)24 * N, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL((k_linearize_t<false>), dim3((F + TPB - 1) / TPB), dim3(TPB), 0, s, 0, F, (const int *)nullptr, gp.d_fa.p, gp.d_fb.p, gp.d_z.p, gp.d_W.p,
                       gp.d_lp.p, gp.d_state.p, c.d_swap.p, c.dp.slot_blk, c.dp.slot_rhs, c.d_H.p, c.d_bad.p, (const double *)nullptr);
    if (what == 0) {
        std::vector<double> H((size_t)9 * std::max(1, P.n_slots));
        HIPCHECK(hipMemcpyAsync(H.data(), c.d_H.p, H.size() * 8, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
        for (int f = 0; f < F; f++) {
            double *o = out + (size_t)33 * f;
            memset(o, 0, 33 * 8);
            memcpy(o, &H[(size_t)9 * P.slot_blk[3 * f]], 72);
            memcpy(o + 27, &H[(size_t)9 * P.slot_rhs[2 * f]], 24);
            if (gp.h_fb.p[f] < 0) continue;
            const double *b1 = &H[(size_t)9 * P.slot_blk[3 * f + 1]];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o[9 + i * 3 + j] = P.fac_swap[f] ? b1[j * 3 + i] : b1[i * 3 + j];
            memcpy(o + 18, &H[(size_t)9 * P.slot_blk[3 * f + 2]], 72);
            memcpy(o + 30, &H[(size_t)9 * P.slot_rhs[2 * f + 1]], 24);
        }
        return 0;
    }
    if (what != 1 || N > 2000) return -2;
    const int nd = (int)P.dest.size();
    DBuf<double> d_out; d_out.need((size_t)9 * std::max(1, nd));
    hipLaunchKernelGGL(k_debug_dest, dim3((9 * nd + TPB - 1) / TPB), dim3(TPB), 0, s, nd, c.dp.dest, c.dp.src_idx, c.d_H.p, d_out.p);
    std::vector<double> D((size_t)9 * std::max(1, nd));
    HIPCHECK(hipMemcpyAsync(D.data(), d_out.p, D.size() * 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    d_out.release();
    const size_t n = (size_t)3 * N;
    double *A = out, *B = out + n * n;
    memset(out, 0, (n * n + n) * 8);
    for (int t = 0; t < P.nF; t++) {
        auto node_of = [&](int lb) { const int pos = lb < P.f_nsb[t] ? P.f_first[t] + lb : P.f_rows[P.f_rows_ptr[t] + lb - P.f_nsb[t]]; return P.perm[pos]; };
        for (int d = P.dest_front_ptr[t]; d < P.dest_front_ptr[t + 1]; d++) {
            const Plan::DestRec &r = P.dest[d];
            const double *v = &D[(size_t)9 * d];
            const int nc = node_of(r.bcol);
            if (r.brow < 0) { for (int j = 0; j < 3; j++) B[(size_t)3 * nc + j] += v[j]; continue; }
            const int nr = node_of(r.brow);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    if (r.brow == r.bcol && i < j) continue;          // diagonal blocks: the assembly takes the lower part
                    const size_t rr = (size_t)3 * nr + i, cc = (size_t)3 * nc + j;
                    A[rr * n + cc] += v[i * 3 + j];
                    if (rr != cc) A[cc * n + rr] += v[i * 3 + j];
                }
        }
    }
    for (size_t i = 0; i < n; i++) A[i * n + i] += param->tikhanov > 0 ? param->tikhanov : 0.0;      // aprilsam.c:197-204
    return 0;
}

// debug: copy the per-front clock stamps (8 per front) written when APRILSAM_AMD_KPROF is set
int debug_front_times(const april_graph_cholesky_param_t *param, long long *out, int n_fronts) {
    return guarded_rc(param, nullptr, [&]() -> int {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_ctx.find(param);
        if (it == g_ctx.end() || !it->second->d_prof.p) return -1;
        HIPCHECK(hipDeviceSynchronize());
        int n = std::min(n_fronts, it->second->plan.nF);
        HIPCHECK(hipMemcpy(out, it->second->d_prof.p, (size_t)8 * PROF_SLOTS * n, hipMemcpyDeviceToHost));
        return n;
    });
}
// ------------------------------------------------------------------------------------------------------
// multi-GPU: nested-dissection subtree sharding (BASELINE.json config 5)
//
// Every rank builds the SAME plan (the planner is deterministic).  The assembly tree is split by proportional
// mapping: the root owns the rank range [0, world); a front with range [lo, hi) is owned by rank lo and hands the
// halves [lo, mid) / [mid, hi) to its children, greedily balanced by subtree flops; ranges of size 1 make a whole
// subtree local.  Per Gauss-Newton iteration the only data crossing ranks are
//   * up:   the Schur update block of a front whose parent lives on another rank (the lower trapezoid of columns
//           3*nsb.. end of its frontal array, packed by k_pack_update), sent point-to-point to the parent's owner,
//   * down: the solved x of the "top" fronts (range > 1 rank), a few thousand doubles each, broadcast.
// The exchange happens inside the library (shard_iterate), over one of the two transports below; aprilsam_amd/shard.py
// is only a launcher (one process per GPU) that hands the RCCL unique id / the host callbacks over.
// ------------------------------------------------------------------------------------------------------
// ---- transports -----------------------------------------------------------------------------------------------
// RCCL (librccl.so, loaded at run time: point-to-point send / recv of the Schur slabs, broadcast of the separator
// solutions, all enqueued on the solver's own HIP stream -- no host synchronisation between a level's kernels and its
// exchange) or host callbacks (the caller moves pinned host buffers with whatever it has: the tests use gloo, a C host
// could use MPI); the schedule above them is the same.
struct Transport {
    bool failed = false; std::string error;       // a communication error ends the sharded run with a return code, not the process
    virtual ~Transport() {}
    virtual void group_begin() {}
    virtual void group_end() {}
    virtual void send(const double *dev, long long n, int dst, hipStream_t s) = 0;
    virtual void recv(double *dev, long long n, int src, hipStream_t s) = 0;
    virtual void bcast(double *dev, long long n, int root, hipStream_t s) = 0;
    virtual void allreduce_sum(double *dev, long long n, hipStream_t s) = 0;
    virtual const char *name() const = 0;
};

struct RcclApi {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr; decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr; decltype(&ncclSend) Send = nullptr; decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr; decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr; decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr; decltype(&ncclCommUserRank) CommUserRank = nullptr; decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string path;
    bool load() {
        if (h) return true;
        // the RCCL that belongs to the HIP runtime THIS library runs on: same directory as the libamdhip64 we are linked to.
        // (A process may hold a second ROCm stack -- PyTorch wheels bundle their own libamdhip64 / librccl -- and a
        // communicator created by that one cannot take our streams.)  Plain sonames only as a fall-back.
        std::string dir;
        Dl_info di;
        if (dladdr((const void *)&hipStreamSynchronize, &di) && di.dli_fname) { dir = di.dli_fname; const size_t k = dir.rfind('/'); dir = k == std::string::npos ? "" : dir.substr(0, k + 1); }
        const std::string cand[] = { dir + "librccl.so.1", dir + "librccl.so", "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so" };
        for (const std::string &nm : cand) { if (nm.empty()) continue; h = dlopen(nm.c_str(), RTLD_NOW | RTLD_LOCAL); if (h) { path = nm; break; } }
        if (!h) return false;
#define RCCL_SYM(x) x = (decltype(x))dlsym(h, "nccl" #x); if (!x) return false
        RCCL_SYM(GetUniqueId); RCCL_SYM(CommInitRank); RCCL_SYM(CommDestroy); RCCL_SYM(Send); RCCL_SYM(Recv); RCCL_SYM(Broadcast);
        RCCL_SYM(AllReduce); RCCL_SYM(GroupStart); RCCL_SYM(GroupEnd); RCCL_SYM(GetErrorString);
        RCCL_SYM(CommCount); RCCL_SYM(CommUserRank); RCCL_SYM(GetVersion);
#undef RCCL_SYM
        return true;
    }
};
static RcclApi g_rccl;
struct RcclTransport : Transport {
    ncclComm_t comm = nullptr;
    ~RcclTransport() override { if (comm) (void)g_rccl.CommDestroy(comm); }
    void chk(ncclResult_t r, const char *what) {
        if (r == ncclSuccess || failed) return;
        failed = true; error = std::string(what) + ": " + g_rccl.GetErrorString(r);
        fprintf(stderr, "aprilsam_amd: RCCL error in %s\n", error.c_str());
    }
    void group_begin() override { if (!failed) chk(g_rccl.GroupStart(), "ncclGroupStart"); }
    void group_end() override { if (!failed) chk(g_rccl.GroupEnd(), "ncclGroupEnd"); }
    void send(const double *dev, long long n, int dst, hipStream_t s) override { if (!failed) chk(g_rccl.Send(dev, (size_t)n, ncclFloat64, dst, comm, s), "ncclSend"); }
    void recv(double *dev, long long n, int src, hipStream_t s) override { if (!failed) chk(g_rccl.Recv(dev, (size_t)n, ncclFloat64, src, comm, s), "ncclRecv"); }
    void bcast(double *dev, long long n, int root, hipStream_t s) override { if (!failed) chk(g_rccl.Broadcast(dev, dev, (size_t)n, ncclFloat64, root, comm, s), "ncclBroadcast"); }
    void allreduce_sum(double *dev, long long n, hipStream_t s) override { if (!failed) chk(g_rccl.AllReduce(dev, dev, (size_t)n, ncclFloat64, ncclSum, comm, s), "ncclAllReduce"); }
    const char *name() const override { return "rccl"; }
};

struct HostTransport : Transport {
    aprilsam_amd_host_comm_t cb{};
    HBuf<double> stage;
    void down(const double *dev, long long n, hipStream_t s) { stage.need((size_t)n); HIPCHECK(hipMemcpyAsync(stage.p, dev, (size_t)n * 8, hipMemcpyDeviceToHost, s)); HIPCHECK(hipStreamSynchronize(s)); }
    void up(double *dev, long long n, hipStream_t s) { HIPCHECK(hipMemcpyAsync(dev, stage.p, (size_t)n * 8, hipMemcpyHostToDevice, s)); HIPCHECK(hipStreamSynchronize(s)); }
    void chk(int rc, const char *what) {
        if (rc == 0 || failed) return;
        failed = true; error = std::string("host communication callback ") + what + " returned " + std::to_string(rc);
        fprintf(stderr, "aprilsam_amd: %s\n", error.c_str());
    }
    void send(const double *dev, long long n, int dst, hipStream_t s) override { if (failed) return; down(dev, n, s); chk(cb.send(cb.user, stage.p, n, dst), "send"); }
    void recv(double *dev, long long n, int src, hipStream_t s) override { if (failed) return; stage.need((size_t)n); chk(cb.recv(cb.user, stage.p, n, src), "recv"); up(dev, n, s); }
    void bcast(double *dev, long long n, int root, hipStream_t s) override { if (failed) return; down(dev, n, s); chk(cb.bcast(cb.user, stage.p, n, root), "bcast"); up(dev, n, s); }
    void allreduce_sum(double *dev, long long n, hipStream_t s) override { if (failed) return; down(dev, n, s); chk(cb.allreduce_sum(cb.user, stage.p, n), "allreduce_sum"); up(dev, n, s); }
    const char *name() const override { return "host callbacks"; }
    ~HostTransport() override { stage.release(); }
};

struct ShardState {
    int rank = 0, world = 1;
    std::vector<int> owner;                  // per front
    std::vector<char> top;                   // per front: rank range spans more than one rank
    ShardLayout lay;                         // this rank's front pool: owned fronts + ghosts of remote children
    std::vector<LevelPlan> levels;           // launch tables of the fronts THIS rank owns, per level
    DBuf<int> d_tab;                         // ... their device copy
    DBuf<int> d_flist; int n_flist = 0;      // factors owned by this rank's fronts
    DBuf<int> d_nown;                        // per node: rank that owns the front eliminating it
    std::vector<long long> xfer;             // transfers up: level, front, src, dst, offset (doubles), count (doubles)
    std::vector<long long> bcast;            // broadcasts down: level, front, owner, first position, own blocks
    struct Xfer { int front, src, dst; long long count, boff; };
    std::vector<std::vector<Xfer>> up;       // per level, global list order; boff = offset in d_send (src == rank) / d_recv (dst == rank)
    std::vector<std::vector<std::array<long long, 3>>> down;   // per level: owner, 3 * first, 3 * nsb
    DBuf<double> d_send, d_recv, d_scratch;
    std::unique_ptr<Transport> tr;
    void release() { d_tab.release(); d_flist.release(); d_nown.release(); d_send.release(); d_recv.release(); d_scratch.release(); tr.reset(); }
};
static std::unordered_map<const void *, std::unique_ptr<ShardState>> g_shard;
static void drop_shard_state(const void *param) {           // (failure path; g_mu held)
    auto it = g_shard.find(param);
    if (it != g_shard.end()) { it->second->release(); g_shard.erase(it); }
}

// Ownership of the fronts and the exchange lists of a `world`-rank run: pure host logic on the plan (also reachable
// without a GPU through aprilsam_amd_shard_plan, tests/test_distributed_cpu.py)
void shard_map(const Plan &P, int world, std::vector<int> &owner, std::vector<char> &top, std::vector<long long> &xfer, std::vector<long long> &bcast) {
    struct { std::vector<int> &owner; std::vector<char> &top; std::vector<long long> &xfer, &bcast; } S{ owner, top, xfer, bcast };
    S.xfer.clear(); S.bcast.clear();
    // subtree work (flops proxy) per front
    std::vector<double> work(P.nF, 0.0);
    for (int t = 0; t < P.nF; t++) {
        const double ns = 3.0 * P.f_nsb[t], m = 3.0 * (P.f_nsb[t] + P.f_nub[t]);
        work[t] += ns * m * m + 1.0;
        if (P.f_parent[t] >= 0) work[P.f_parent[t]] += work[t];
    }
    std::vector<int> lo(P.nF, 0), hi(P.nF, world);
    S.owner.assign(P.nF, 0); S.top.assign(P.nF, 0);
    // roots first (fronts are numbered children-before-parents, so walk downwards from the end)
    std::vector<std::vector<int>> kids(P.nF);
    for (int t = 0; t < P.nF; t++) if (P.f_parent[t] >= 0) kids[P.f_parent[t]].push_back(t);
    {   // several roots (disconnected graph): spread them like children of a virtual root
        std::vector<int> roots; for (int t = 0; t < P.nF; t++) if (P.f_parent[t] < 0) roots.push_back(t);
        std::sort(roots.begin(), roots.end(), [&](int a, int b) { return work[a] != work[b] ? work[a] > work[b] : a < b; });
        for (size_t i = 0; i < roots.size(); i++) { lo[roots[i]] = 0; hi[roots[i]] = world; }       // (every root spans all ranks)
    }
    for (int t = P.nF - 1; t >= 0; t--) {
        S.owner[t] = lo[t]; S.top[t] = (hi[t] - lo[t]) > 1;
        if (kids[t].empty()) continue;
        if (hi[t] - lo[t] <= 1) { for (int ch : kids[t]) { lo[ch] = lo[t]; hi[ch] = hi[t]; } continue; }
        const int mid = (lo[t] + hi[t]) / 2;
        std::vector<int> ks(kids[t]);
        std::sort(ks.begin(), ks.end(), [&](int a, int b) { return work[a] != work[b] ? work[a] > work[b] : a < b; });
        double wa = 0, wb = 0;
        for (int ch : ks) {
            if (wa <= wb) { wa += work[ch]; lo[ch] = lo[t]; hi[ch] = mid; }
            else { wb += work[ch]; lo[ch] = mid; hi[ch] = hi[t]; }
        }
    }
    // exchange lists
    for (int t = 0; t < P.nF; t++) {
        const int par = P.f_parent[t];
        if (par >= 0 && S.owner[par] != S.owner[t]) {
            const long long R = P.rows(t), C = P.cols(t), ns = 3ll * P.f_nsb[t];
            const long long v[6] = { P.f_level[t], t, S.owner[t], S.owner[par], P.f_off[t], upd_packed_offset((int)R, (int)ns, (int)C) };
            S.xfer.insert(S.xfer.end(), v, v + 6);
        }
        if (S.top[t]) { const long long v[5] = { P.f_level[t], t, S.owner[t], P.f_first[t], P.f_nsb[t] }; S.bcast.insert(S.bcast.end(), v, v + 5); }
    }
}

// modelled critical path of a mapping, in sum c_j^2 flops.  Fronts whose rank range spans more than one rank ("top" fronts) each
// run on ONE owner; those on different branches run side by side, those on one root path one after the other: the serial
// part is the heaviest root path through the top fronts.  Below them every rank works through its own subtrees in parallel.
// {whole factorisation, heaviest root path of top fronts, busiest rank's subtrees, all top fronts together}
std::vector<long long> shard_critical_path(const Plan &P, int world, const std::vector<int> &owner, const std::vector<char> &top) {
    double total = 0, topall = 0, path_max = 0; std::vector<double> local(world, 0.0), path(P.nF, 0.0);
    for (int t = 0; t < P.nF; t++) {                    // (fronts are numbered children before parents)
        const double ns = 3.0 * P.f_nsb[t], nu = 3.0 * P.f_nub[t];
        double fl = 0; for (int q = 0; q < (int)ns; q++) { const double cj = (ns - q) + nu + 1; fl += cj * cj; }
        total += fl;
        if (!top[t]) { local[owner[t]] += fl; continue; }
        topall += fl; path[t] += fl;
        path_max = std::max(path_max, path[t]);
        if (P.f_parent[t] >= 0) path[P.f_parent[t]] = std::max(path[P.f_parent[t]], path[t]);
    }
    return { (long long)total, (long long)path_max, (long long)*std::max_element(local.begin(), local.end()), (long long)topall };
}

// Every rank calls this with the same graph.  Builds the (identical) plan, the ownership map, THIS rank's pool layout
// (owned fronts + ghosts), launch tables and exchange buffers.  A transport must be attached before the first
// iteration unless world == 1 (shard_comm_init_rccl / shard_comm_init_host).
static int shard_begin_impl(april_graph_t *g, april_graph_cholesky_param_t *param, int rank, int world);
int shard_begin(april_graph_t *g, april_graph_cholesky_param_t *param, int rank, int world) { return guarded_rc(param, g, [&] { return shard_begin_impl(g, param, rank, world); }); }
static int shard_begin_impl(april_graph_t *g, april_graph_cholesky_param_t *param, int rank, int world) {
    if (zsize(g->nodes) == 0 || zsize(g->factors) == 0 || world < 1 || rank < 0 || rank >= world) return -1;
    ensure_device();
    std::lock_guard<std::mutex> lk(g_mu);
    Context &c = ctx_for(param);
    GraphPack &gp = pack_for(g);
    pack_factors(gp, g);
    if (!gp.host_idx.empty()) return -4;
    pack_states(gp, g, false);
    c.have_plan = false;                      // the pool layout is per rank: never reuse an upload made for another layout
    prepare_plan(c, gp, g, false);
    const Plan &P = c.plan;
    { auto it = g_shard.find(param); if (it != g_shard.end()) { it->second->release(); g_shard.erase(it); } }
    auto &S = *(g_shard[param] = std::make_unique<ShardState>());
    S.rank = rank; S.world = world;
    shard_map(P, world, S.owner, S.top, S.xfer, S.bcast);
    // ---- pool layout: owned fronts in plan order, then the ghosts of remote children -------------------------------
    S.lay.off.assign(P.nF, -1); S.lay.ghost.assign(P.nF, 0);
    long long run = 0;
    for (int t = 0; t < P.nF; t++) if (S.owner[t] == rank) { run = (run + 31) & ~31ll; S.lay.off[t] = run; run += (long long)P.rows(t) * P.cols(t); }
    for (int t = 0; t < P.nF; t++) {
        const int par = P.f_parent[t];
        if (par >= 0 && S.owner[par] == rank && S.owner[t] != rank) {
            run = (run + 31) & ~31ll; S.lay.off[t] = run; S.lay.ghost[t] = 1;
            run += (long long)(3 * P.f_nub[t] + 3) * (3 * P.f_nub[t]);
        }
    }
    S.lay.pool_doubles = run;
    upload_plan(c, gp.stream, &S.lay);
    c.have_plan = false;                      // (a later non-sharded call on this param must re-upload the full layout)
    upload_factors(gp);
    set_lambda(c, gp, param->tikhanov);
    // ---- launch tables of the owned fronts ------------------------------------------------------------------------
    std::vector<int> tab;
    S.levels.assign(P.nLevels, LevelPlan());
    for (int l = 0; l < P.nLevels; l++) {
        std::vector<int> fr;
        for (int k = P.lev_ptr[l]; k < P.lev_ptr[l + 1]; k++) if (S.owner[P.lev_fronts[k]] == rank) fr.push_back(P.lev_fronts[k]);
        build_level(S.levels[l], fr, tab, [&](int t, int *nsb, int *nub) { *nsb = P.f_nsb[t]; *nub = P.f_nub[t]; }, [&](int t) { return c.inc.fd[t].dinv0 >= 0; });
    }
    if (tab.empty()) tab.push_back(0);
    S.d_tab.need(tab.size());
    HIPCHECK(hipMemcpyAsync(S.d_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, gp.stream));
    { size_t mx = 1; for (int l = 0; l < P.nLevels; l++) mx = std::max(mx, diag_doubles(S.levels[l].n_big, S.levels[l].n_diag_slots)); c.d_diag.need(mx); }
    // ---- factors owned by this rank's fronts, node ownership ----------------------------------------------------
    std::vector<int> fl;
    for (int f = 0; f < P.F; f++) if (P.fac_front[f] >= 0 && S.owner[P.fac_front[f]] == rank) fl.push_back(f);
    S.n_flist = (int)fl.size();
    S.d_flist.need(std::max<size_t>(1, fl.size()));
    if (!fl.empty()) HIPCHECK(hipMemcpyAsync(S.d_flist.p, fl.data(), fl.size() * 4, hipMemcpyHostToDevice, gp.stream));
    std::vector<int> nown(P.N, 0);
    for (int t = 0; t < P.nF; t++) for (int k = 0; k < P.f_nsb[t]; k++) nown[P.perm[P.f_first[t] + k]] = S.owner[t];
    S.d_nown.need(P.N);
    HIPCHECK(hipMemcpyAsync(S.d_nown.p, nown.data(), (size_t)P.N * 4, hipMemcpyHostToDevice, gp.stream));
    // ---- exchange lists of this rank ------------------------------------------------------------------------------
    S.up.assign(P.nLevels, {}); S.down.assign(P.nLevels, {});
    long long send_max = 1, recv_max = 1;
    for (size_t i = 0; i + 6 <= S.xfer.size(); i += 6) {
        const int lev = (int)S.xfer[i], front = (int)S.xfer[i + 1], src = (int)S.xfer[i + 2], dst = (int)S.xfer[i + 3];
        S.up[lev].push_back({ front, src, dst, S.xfer[i + 5], 0 });
    }
    for (int l = 0; l < P.nLevels; l++) {
        long long so = 0, ro = 0;
        for (auto &x : S.up[l]) {
            if (x.src == rank) { x.boff = so; so += x.count; }
            else if (x.dst == rank) { x.boff = ro; ro += x.count; }
        }
        send_max = std::max(send_max, so); recv_max = std::max(recv_max, ro);
    }
    for (size_t i = 0; i < S.bcast.size(); i += 5) S.down[(int)S.bcast[i]].push_back({ S.bcast[i + 2], 3 * S.bcast[i + 3], 3 * S.bcast[i + 4] });
    S.d_send.need((size_t)send_max); S.d_recv.need((size_t)recv_max);
    HIPCHECK(hipMemsetAsync(c.d_x.p, 0, (size_t)24 * gp.N, gp.stream));     // poses of other ranks' subtrees simply do not move here
    HIPCHECK(hipStreamSynchronize(gp.stream));      // host vectors above go out of scope
    c.st.n_nodes = gp.N; c.st.n_factors = gp.F;
    return 0;
}
// what: 0 -> {levels, fronts, nodes, pool doubles of this rank, pool doubles of the whole plan}, 1 -> xfer (6 per entry),
// 2 -> bcast (5 per entry), 3 -> owner per front.  Returns count written (or needed if out == null)
long long shard_info(const april_graph_cholesky_param_t *param, int what, long long *out, long long cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_shard.find(param); auto ic = g_ctx.find(param);
    if (it == g_shard.end() || ic == g_ctx.end()) return -1;
    const ShardState &S = *it->second;
    std::vector<long long> v;
    if (what == 0) v = { ic->second->plan.nLevels, ic->second->plan.nF, ic->second->plan.N, S.lay.pool_doubles, (long long)ic->second->plan.pool_doubles };
    else if (what == 1) v = S.xfer;
    else if (what == 2) v = S.bcast;
    else if (what == 3) v.assign(S.owner.begin(), S.owner.end());
    else if (what == 4) v = shard_critical_path(ic->second->plan, S.world, S.owner, S.top);
    if (out) for (long long i = 0; i < (long long)v.size() && i < cap; i++) out[i] = v[i];
    return (long long)v.size();
}

// ---- attaching a transport --------------------------------------------------------------------------------------
int shard_comm_unique_id(char *out128) {
    const int dev_rc = guarded_rc(nullptr, nullptr, [&] { ensure_device(); return 0; });
    if (dev_rc) return dev_rc;
    if (!g_rccl.load()) return -5;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { fprintf(stderr, "aprilsam_amd: ncclGetUniqueId failed: %s\n", g_rccl.GetErrorString(r)); return -6; }
    memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}
static int shard_comm_init_rccl_impl(const april_graph_cholesky_param_t *param, const char *id128);
int shard_comm_init_rccl(const april_graph_cholesky_param_t *param, const char *id128) { return guarded_rc(param, nullptr, [&] { return shard_comm_init_rccl_impl(param, id128); }); }
static int shard_comm_init_rccl_impl(const april_graph_cholesky_param_t *param, const char *id128) {
    ensure_device();
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_shard.find(param);
    if (it == g_shard.end()) return -1;
    if (!g_rccl.load()) return -5;
    ShardState &S = *it->second;
    auto T = std::make_unique<RcclTransport>();
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    HIPCHECK(hipSetDevice(g_device));
    const ncclResult_t r = g_rccl.CommInitRank(&T->comm, S.world, id, S.rank);
    if (r != ncclSuccess) { fprintf(stderr, "aprilsam_amd: ncclCommInitRank failed: %s\n", g_rccl.GetErrorString(r)); return -6; }
    S.tr = std::move(T);
    return 0;
}
// what the attached transport is, as the communication library itself reports it: out = {kind (0 none, 1 RCCL, 2 host
// callbacks), ncclCommCount, ncclCommUserRank, ncclGetVersion code, HIP device}; path (may be null) receives the librccl
// file the symbols came from
int shard_comm_info(const april_graph_cholesky_param_t *param, long long *out, char *path, int cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_shard.find(param);
    if (it == g_shard.end()) return -1;
    ShardState &S = *it->second;
    out[0] = 0; out[1] = S.world; out[2] = S.rank; out[3] = 0; out[4] = g_device;
    if (path && cap > 0) path[0] = 0;
    if (!S.tr) return 0;
    if (auto *R = dynamic_cast<RcclTransport *>(S.tr.get())) {
        int cnt = -1, ur = -1, ver = 0;
        (void)g_rccl.CommCount(R->comm, &cnt); (void)g_rccl.CommUserRank(R->comm, &ur); (void)g_rccl.GetVersion(&ver);
        out[0] = 1; out[1] = cnt; out[2] = ur; out[3] = ver;
        if (path && cap > 0) { strncpy(path, g_rccl.path.c_str(), (size_t)cap - 1); path[cap - 1] = 0; }
    } else out[0] = 2;
    return 0;
}
int shard_comm_init_host(const april_graph_cholesky_param_t *param, const aprilsam_amd_host_comm_t *cb) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_shard.find(param);
    if (it == g_shard.end() || !cb || !cb->send || !cb->recv || !cb->bcast || !cb->allreduce_sum) return -1;
    auto T = std::make_unique<HostTransport>();
    T->cb = *cb;
    it->second->tr = std::move(T);
    return 0;
}

// n Gauss-Newton iterations of the sharded solve: per level the owned fronts, then the Schur slabs whose parent lives on
// another rank (packed lower trapezoid, point to point); on the way down the solved x of the top fronts (broadcast).
static int shard_iterate_impl(april_graph_t *g, april_graph_cholesky_param_t *param, int n);
int shard_iterate(april_graph_t *g, april_graph_cholesky_param_t *param, int n) { return guarded_rc(param, g, [&] { return shard_iterate_impl(g, param, n); }); }
static int shard_iterate_impl(april_graph_t *g, april_graph_cholesky_param_t *param, int n) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_shard.find(param); auto ic = g_ctx.find(param);
    if (it == g_shard.end() || ic == g_ctx.end()) return -1;
    ShardState &S = *it->second; Context &c = *ic->second;
    if (S.world > 1 && !S.tr) return -7;
    GraphPack &gp = pack_for(g);
    const Plan &P = c.plan;
    hipStream_t s = gp.stream;
    HIPCHECK(hipSetDevice(g_device));
    set_small_attr();
    const int N = gp.N, me = S.rank;
    auto nop = [](int) {}; auto nop0 = []() {};
    Transport *T = S.tr.get();
    gp.mirror_sync = false;
    HIPCHECK(hipMemsetAsync(c.d_bad.p, 0, 16, s));          // sticky over the n iterations: the first failure is the one reported
    for (int iter = 0; iter < n; iter++) {
        HIPCHECK(hipMemcpyAsync(gp.d_lp.p, gp.d_state.p, (size_t)24 * N, hipMemcpyDeviceToDevice, s));      // relinearise
        if (S.n_flist)
            hipLaunchKernelGGL((k_linearize_t<false>), dim3((S.n_flist + TPB - 1) / TPB), dim3(TPB), 0, s, 0, S.n_flist, (const int *)S.d_flist.p, gp.d_fa.p, gp.d_fb.p,
                               gp.d_z.p, gp.d_W.p, gp.d_lp.p, gp.d_state.p, c.d_swap.p, c.dp.slot_blk, c.dp.slot_rhs, c.d_H.p, (int *)nullptr, (const double *)nullptr);
        for (int l = 0; l < P.nLevels; l++) {
            c.la_next = 0;
            enqueue_factor_level(c, S.levels[l], s, nop, nop0, g_opt.lookahead != 0, S.d_tab.p);
            if (S.up[l].empty() || !T) continue;
            bool any = false;
            for (const auto &x : S.up[l]) {
                if (x.src != me) continue;
                const int R = P.rows(x.front), C = P.cols(x.front), ns = 3 * P.f_nsb[x.front];
                if (C > ns) hipLaunchKernelGGL(k_pack_update, dim3(C - ns), dim3(TPB), 0, s, c.d_pool.p + S.lay.off[x.front], R, ns, S.d_send.p + x.boff, 0);
                any = true;
            }
            for (const auto &x : S.up[l]) any = any || x.dst == me;
            if (!any) continue;
            T->group_begin();
            for (const auto &x : S.up[l]) {
                if (x.src == me) T->send(S.d_send.p + x.boff, x.count, x.dst, s);
                else if (x.dst == me) T->recv(S.d_recv.p + x.boff, x.count, x.src, s);
            }
            T->group_end();
            for (const auto &x : S.up[l]) {
                if (x.dst != me) continue;
                const int cnu = P.f_nub[x.front];       // ghost = the update block alone: a front with no own columns
                if (cnu > 0) hipLaunchKernelGGL(k_pack_update, dim3(3 * cnu), dim3(TPB), 0, s, c.d_pool.p + S.lay.off[x.front], 3 * cnu + 3, 0, S.d_recv.p + x.boff, 1);
            }
        }
        for (int l = P.nLevels - 1; l >= 0; l--) {
            launch_backsolve(c, S.levels[l], s, nop, nop0, S.d_tab.p);
            if (S.down[l].empty() || !T) continue;
            T->group_begin();
            for (const auto &b : S.down[l]) T->bcast(c.d_x.p + b[1], b[2], (int)b[0], s);
            T->group_end();
        }
        hipLaunchKernelGGL(k_update_states, dim3((N + TPB - 1) / TPB), dim3(TPB), 0, s, N, c.d_pos.p, c.d_x.p, gp.d_lp.p, gp.d_state.p, gp.d_dx.p,
                           (double *)nullptr, (double *)nullptr, (const int *)nullptr, (int *)nullptr);
        HIPCHECK(hipGetLastError());
        if (T && T->failed) break;
    }
    HIPCHECK(hipMemcpyAsync(c.h_bad.p, c.d_bad.p, 16, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (T && T->failed) return -6;
    // a pivot fails on ONE rank (the owner of the front); every rank must leave with the same answer, or the others walk
    // into the next collective alone: the flags are added up over the transport
    double flag = c.h_bad.p[0] ? ((c.h_bad.p[0] == 9 || c.h_bad.p[2] == 9) ? 1e6 : 1.0) : 0.0;
    if (T) {
        gp.d_scalar.need(8); gp.h_scalar.need(8);
        gp.h_scalar.p[0] = flag;
        HIPCHECK(hipMemcpyAsync(gp.d_scalar.p, gp.h_scalar.p, 8, hipMemcpyHostToDevice, s));
        HIPCHECK(hipStreamSynchronize(s));
        T->allreduce_sum(gp.d_scalar.p, 1, s);
        HIPCHECK(hipMemcpyAsync(gp.h_scalar.p, gp.d_scalar.p, 8, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
        if (T->failed) return -6;
        flag = gp.h_scalar.p[0];
    }
    if (flag >= 1e6) fail(ERR_DEP_TIMEOUT, "sharded solve: a multi-level launch gave up waiting for a dependency flag");
    c.st.not_spd = flag != 0;
    return flag != 0 ? -2 : 0;
}

// After the iterations every rank holds the states of its own subtrees and of the top fronts.  Gather: states, l_points
// and dx masked by node ownership, summed over the ranks (x + 0 + ... + 0 is exact: every rank ends up with bit-identical
// copies), written into the device arrays and into the caller's node objects like a resident run does.
static int shard_gather_states_impl(april_graph_t *g, april_graph_cholesky_param_t *param);
int shard_gather_states(april_graph_t *g, april_graph_cholesky_param_t *param) { return guarded_rc(param, g, [&] { return shard_gather_states_impl(g, param); }); }
static int shard_gather_states_impl(april_graph_t *g, april_graph_cholesky_param_t *param) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_shard.find(param); auto ic = g_ctx.find(param);
    if (it == g_shard.end() || ic == g_ctx.end()) return -1;
    ShardState &S = *it->second; Context &c = *ic->second;
    if (S.world > 1 && !S.tr) return -7;
    GraphPack &gp = pack_for(g);
    hipStream_t s = gp.stream;
    HIPCHECK(hipSetDevice(g_device));
    const int N = gp.N;
    if (S.tr) {
        S.d_scratch.need((size_t)9 * N);
        hipLaunchKernelGGL(k_mask_owned, dim3((3 * N + TPB - 1) / TPB), dim3(TPB), 0, s, 3 * N, S.d_nown.p, S.rank, gp.d_state.p, gp.d_lp.p, gp.d_dx.p, S.d_scratch.p);
        S.tr->allreduce_sum(S.d_scratch.p, (long long)9 * N, s);
        if (S.tr->failed) { HIPCHECK(hipStreamSynchronize(s)); return -6; }
        HIPCHECK(hipMemcpyAsync(gp.d_state.p, S.d_scratch.p, (size_t)24 * N, hipMemcpyDeviceToDevice, s));
        HIPCHECK(hipMemcpyAsync(gp.d_lp.p, S.d_scratch.p + (size_t)3 * N, (size_t)24 * N, hipMemcpyDeviceToDevice, s));
        HIPCHECK(hipMemcpyAsync(gp.d_dx.p, S.d_scratch.p + (size_t)6 * N, (size_t)24 * N, hipMemcpyDeviceToDevice, s));
    }
    HIPCHECK(hipMemcpyAsync(gp.h_state.p, gp.d_state.p, (size_t)24 * N, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(gp.h_lp.p, gp.d_lp.p, (size_t)24 * N, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipMemcpyAsync(gp.h_dx.p, gp.d_dx.p, (size_t)24 * N, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    for (int i = 0; i < N; i++) {
        april_graph_node_t *nd = ns[i];
        nd->UID = i;
        memcpy(nd->state, gp.h_state.p + (size_t)3 * i, 24);
        memcpy(nd->l_point, gp.h_lp.p + (size_t)3 * i, 24);
        const double *dx = gp.h_dx.p + (size_t)3 * i;
        if (!(std::isnan(dx[0]) || std::isnan(dx[1]) || std::isnan(dx[2]))) memcpy(nd->delta_X, dx, 24);
    }
    if (param->ordering) free(param->ordering);
    param->ordering = (int *)malloc(sizeof(int) * (size_t)N);
    memcpy(param->ordering, c.plan.perm.data(), sizeof(int) * (size_t)N);
    param->nreordering = N; param->factor_num = gp.Fg;
    return 0;
}
// chi^2 at the resident states: every rank sums the factors its fronts own, the transport adds the partial sums
static double shard_chi2_impl(april_graph_t *g, april_graph_cholesky_param_t *param);
double shard_chi2(april_graph_t *g, april_graph_cholesky_param_t *param) {
    double out = std::nan("");
    guarded(param, g, [&] { out = shard_chi2_impl(g, param); });
    return out;
}
static double shard_chi2_impl(april_graph_t *g, april_graph_cholesky_param_t *param) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_shard.find(param); auto ic = g_ctx.find(param);
    if (it == g_shard.end() || ic == g_ctx.end()) return -1;
    GraphPack &gp = pack_for(g);
    const Plan &P = ic->second->plan; ShardState &S = *it->second;
    hipStream_t s = gp.stream;
    HIPCHECK(hipSetDevice(g_device));
    hipLaunchKernelGGL(k_chi2, dim3((gp.F + TPB - 1) / TPB), dim3(TPB), 0, s, gp.F, gp.d_fa.p, gp.d_fb.p, gp.d_z.p, gp.d_W.p, gp.d_state.p, gp.d_chi2f.p);
    std::vector<double> h(gp.F);
    HIPCHECK(hipMemcpyAsync(h.data(), gp.d_chi2f.p, (size_t)8 * gp.F, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    double acc = 0;
    for (int f = 0; f < gp.F; f++) if (P.fac_front[f] >= 0 && S.owner[P.fac_front[f]] == S.rank) acc += h[f];
    if (S.tr) {
        HIPCHECK(hipMemcpyAsync(gp.d_scalar.p, &acc, 8, hipMemcpyHostToDevice, s));
        HIPCHECK(hipStreamSynchronize(s));
        S.tr->allreduce_sum(gp.d_scalar.p, 1, s);
        HIPCHECK(hipMemcpyAsync(gp.h_scalar.p, gp.d_scalar.p, 8, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
        acc = gp.h_scalar.p[0];
    }
    return acc;
}
void shard_end(const april_graph_cholesky_param_t *param) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_shard.find(param);
    if (it != g_shard.end()) { it->second->release(); g_shard.erase(it); }
}

// ------------------------------------------------------------------------------------------------------
// Host-side consistency checks of the index arithmetic that host launch tables and device kernels share
// (kernels.hip.h: syrk_range / syrk_tiles / trapezoid decode, upd_packed_offset, panel_tiles).  No GPU needed.
// Returns 0, or a negative code naming the first failed check.
// ------------------------------------------------------------------------------------------------------
int selftest() {
    // (1) outer-blocked trailing update: replay the launch sequence of one big front on the host and count, for
    //     every element (i >= j) of the front, which panel columns k have been applied when it is consumed.
    const int cases[][2] = { { 12, 0 }, { 12, 30 }, { 45, 17 }, { 100, 0 }, { 130, 64 }, { 43, 200 }, { 11, 11 }, { 87, 1 } };   // (nsb, nub)
    for (auto &cs : cases) {
        const int nsb = cs[0], nub = cs[1], nbc = nsb + nub, R = 3 * (nbc + 1), C = 3 * nbc, ns = 3 * nsb, Rv = R - 2;
        std::vector<int> applied((size_t)Rv * C, 0);          // number of K columns applied to (i, j), j < C, i < Rv
        std::vector<long long> ksum((size_t)Rv * C, 0);       // sum of applied k (detects duplicates / wrong columns)
        const int steps = (ns + NB - 1) / NB;
        auto run = [&](int s_lo, int s_hi, int mode, int tile) -> int {
            const SyrkRange g = syrk_range(R, C, ns, s_lo, s_hi, mode, tile);
            const int nt = syrk_tiles(R, C, ns, s_lo, s_hi, mode, tile);
            if (nt != g.ntc * g.ntr - g.ntc * (g.ntc - 1) / 2) return -11;
            std::vector<char> seen((size_t)std::max(1, g.ntr) * std::max(1, g.ntc), 0);
            for (int l = 0; l < nt; l++) {
                // same decode as trapezoid_tile (host copy of the arithmetic)
                int tj = 0; while (tj + 1 < g.ntc && (tj + 1) * g.ntr - (tj + 1) * tj / 2 <= l) tj++;
                const int ti = tj + (l - (tj * g.ntr - tj * (tj - 1) / 2));
                if (ti < tj || ti >= g.ntr || seen[(size_t)tj * g.ntr + ti]++) return -12;
                for (int j = g.col_lo + tj * tile; j < std::min(g.col_hi, g.col_lo + (tj + 1) * tile); j++)
                    for (int i = std::max(j, g.col_lo + ti * tile); i < std::min(Rv, g.col_lo + (ti + 1) * tile); i++)
                        for (int k = g.k_lo; k < g.k_hi; k++) { applied[(size_t)j * Rv + i]++; ksum[(size_t)j * Rv + i] += k; }
            }
            return 0;
        };
        for (int variant = 0; variant < 4; variant++) {          // wide update whole / split for look-ahead, both tile sizes
            const int tile = (variant & 1) ? TILE2 : TILE; const bool split = variant >= 2;
            std::fill(applied.begin(), applied.end(), 0); std::fill(ksum.begin(), ksum.end(), 0);
            for (int s = 0; s < steps; s++) {
                // when panel s is factored, each of its columns j must carry exactly the columns k < s*NB
                const int k0 = s * NB, k1 = std::min(ns, k0 + NB);
                for (int j = k0; j < k1; j++)
                    for (int i = j; i < Rv; i++)
                        if (applied[(size_t)j * Rv + i] != k0 || ksum[(size_t)j * Rv + i] != (long long)k0 * (k0 - 1) / 2) return -13;
                int rc = run(s, s + 1, 0, TILE); if (rc) return rc;                        // narrow update (always the 64 x 64 kernel)
                if ((s + 1) % OBP == 0 || s + 1 == steps) {
                    if (split) { rc = run(s / OBP * OBP, s + 1, 2, TILE); if (rc) return rc; rc = run(s / OBP * OBP, s + 1, 3, tile); if (rc) return rc; }
                    else { rc = run(s / OBP * OBP, s + 1, 1, tile); if (rc) return rc; }
                }
            }
            for (int j = ns; j < C; j++)
                for (int i = j; i < Rv; i++)
                    if (applied[(size_t)j * Rv + i] != ns || ksum[(size_t)j * Rv + i] != (long long)ns * (ns - 1) / 2) return -14;
        }
        // (2) packed Schur update: offsets are the running count of (rows from the top of the diagonal block to the rhs row)
        long long run_off = 0;
        for (int j = ns; j < C; j++) { if (upd_packed_offset(R, ns, j) != run_off) return -21; run_off += Rv - 3 * (j / 3); }
        if (upd_packed_offset(R, ns, C) != run_off) return -22;
        // (3) row tiles of the panel kernel cover the rows below every panel exactly once
        for (int s = 0; s < steps; s++) {
            const int k0 = s * NB, wdt = std::min(NB, ns - k0), below = Rv - (k0 + wdt);
            const int nt = panel_tiles(R, ns, s);
            if (nt < 1 || (long long)nt * PANEL_ROWS < below || (nt > 1 && (long long)(nt - 1) * PANEL_ROWS >= below)) return -31;
        }
    }
    // (4) LDS budgets: a front classified "full" also fits as "panel", and the work-list region is what the kernel carves
    for (int nw : { 4, 8, 16 })
        for (int R = 6; R < 400; R += 7)
            for (int nsb = 1; 3 * nsb < R - 3; nsb += 5) {
                const int C = R - 3;
                if (panel_front_lds(R, 3 * nsb, nw) > small_front_lds(R, C, nw)) return -41;
                if (small_front_lds(R, C, nw) != (size_t)(R | 1) * C * 8 + (size_t)wl_bytes(nw)) return -42;
            }
    return 0;
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
    const Options before = g_opt;
    if (k == "leaf_nodes") g_opt.leaf_nodes = (int)v;
    else if (k == "deterministic") g_opt.deterministic = (int)v;
    else if (k == "use_graph") g_opt.use_graph = (int)v;
    else if (k == "device_timing") g_opt.device_timing = (int)v;
    else if (k == "trust_factor_cache") g_opt.trust_factor_cache = (int)v;
    else if (k == "small_lds_kb") g_opt.small_lds_kb = (int)v;
    else if (k == "medium_lds_kb") {}                 // accepted for compatibility: the single-workgroup L2 kernel is gone (panel mode)
    else if (k == "syrk128_rows") g_opt.syrk128_rows = (int)v;
    else if (k == "panel_mode") g_opt.panel_mode = (int)v;
    else if (k == "small_threads") g_opt.small_threads = (int)v;
    else if (k == "tp_fronts") g_opt.tp_fronts = (int)v;
    else if (k == "tp_lds_kb") g_opt.tp_lds_kb = (int)v;
    else if (k == "tp_threads") g_opt.tp_threads = (int)v;
    else if (k == "lookahead") g_opt.lookahead = (int)v;
    else if (k == "inc_fast") g_opt.inc_fast = (int)v;
    else if (k == "inc_multi") g_opt.inc_multi = (int)v;
    else if (k == "inc_one") g_opt.inc_one = (int)v;
    else if (k == "inc_one_up") g_opt.inc_one_up = std::max(1, (int)v);
    else if (k == "inc_one_dn") g_opt.inc_one_dn = std::max(1, (int)v);
    else if (k == "inc_one_threads") g_opt.inc_one_threads = (int)v;
    else if (k == "inc_one_spin") g_opt.inc_one_spin = (int)v;
    else if (k == "inc_tail") g_opt.inc_tail = (int)v;
    else if (k == "inc_inline") g_opt.inc_inline = (int)v;
    else if (k == "inc_update") g_opt.inc_update = (int)v;
    else if (k == "inc_tail_solve") g_opt.inc_tail_solve = (int)v;
    else if (k == "inc_lazy_states") g_opt.inc_lazy_states = (int)v;
    else if (k == "inc_replan_tall") g_opt.inc_replan_tall = (int)v;
    else if (k == "speculate_factors") g_opt.speculate_factors = (int)v;
    else if (k == "block_factor") g_opt.block_factor = (int)v;
    else if (k == "pin_last") g_opt.pin_last = (int)v;
    else if (k == "fused_panel") g_opt.fused_panel = (int)v;
    else if (k == "persist") g_opt.persist = (int)v;
    else if (k == "persist_max_fronts") g_opt.persist_max_fronts = (int)v;
    else if (k == "linearize_staged_min") g_opt.linearize_staged_min = (int)v;
    else if (k == "wave_backsolve") g_opt.wave_backsolve = (int)v;
    else if (k == "left_panels") g_opt.left_panels = (int)v;
    else if (k == "block_panels") g_opt.block_panels = (int)v;
    else if (k == "blk_backsolve") g_opt.blk_backsolve = (int)v;
    else if (k == "tile_assembly") g_opt.tile_assembly = (int)v;
    else if (k == "tail_poses") g_opt.tail_poses = std::max(8, (int)v);
    else if (k == "batch_extend") g_opt.batch_extend = (int)v;
    else if (k == "extend_tail_fronts") g_opt.extend_tail_fronts = (int)v;
    else if (k == "mem_cap_mb") g_opt.mem_cap_mb = (int)v;
    else return -1;
    // host-side policies that no launch table or captured graph depends on
    static const char *const no_replan[] = { "deterministic", "use_graph", "device_timing", "trust_factor_cache", "inc_fast", "inc_multi", "inc_one", "inc_one_up", "inc_one_dn", "inc_one_threads", "inc_one_spin", "inc_inline", "inc_update", "inc_tail_solve", "inc_lazy_states", "inc_replan_tall", "speculate_factors", "batch_extend",
                                             "extend_tail_fronts", "mem_cap_mb", "medium_lds_kb" };
    bool policy = false;
    for (const char *q : no_replan) policy = policy || k == q;
    if (!policy && memcmp(&before, &g_opt, sizeof(Options)) != 0) g_opt_epoch++;
    return 0;
}

}  // namespace asam

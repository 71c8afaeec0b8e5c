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
        v = g_opt.syrk_xcd_order; envd("APRILSAM_AMD_SYRK_XCD_ORDER", &v); g_opt.syrk_xcd_order = (int)v;
        v = g_opt.syrk_variant; envd("APRILSAM_AMD_SYRK_VARIANT", &v); g_opt.syrk_variant = (int)v;
        v = g_opt.schur_first; envd("APRILSAM_AMD_SCHUR_FIRST", &v); g_opt.schur_first = (int)v;
        v = g_opt.syrk_small_tiles; envd("APRILSAM_AMD_SYRK_SMALL_TILES", &v); g_opt.syrk_small_tiles = (int)v;
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

// The host runtime continues in the parts below (one translation unit -- the device code of kernels.hip.h is compiled once --
// split by topic):
#include "solver_pack.inc.h"
#include "solver_context.inc.h"
#include "solver_inc.inc.h"
#include "solver_calls.inc.h"
#include "solver_resident.inc.h"
#include "solver_shard.inc.h"

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
        for (int variant = 0; variant < 6; variant++) {          // wide update whole / split for look-ahead, all three tile sizes
            const int tile = variant >= 4 ? TILE / 2 : ((variant & 1) ? TILE2 : TILE); const bool split = variant >= 2 && variant < 4;
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
    // (3b) the XCD-aware tile order of the wide updates is a bijection onto the same trapezoid
    for (int ntr = 1; ntr <= 75; ntr += (ntr < 20 ? 1 : 7))
        for (int ntc = 1; ntc <= ntr; ntc += (ntc < 20 ? 1 : 5)) {
            const int nt = ntc * ntr - ntc * (ntc - 1) / 2;
            std::vector<char> seen((size_t)ntr * ntc, 0);
            for (int l = 0; l < nt; l++) {
                int ti = -1, tj = -1;
                trapezoid_tile_xcd(l, nt, ntr, ntc, &ti, &tj);
                if (tj < 0 || tj >= ntc || ti < tj || ti >= ntr || seen[(size_t)tj * ntr + ti]++) return -35;
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
    else if (k == "syrk_xcd_order") g_opt.syrk_xcd_order = (int)v;
    else if (k == "syrk_variant") g_opt.syrk_variant = (int)v;
    else if (k == "schur_first") g_opt.schur_first = (int)v;
    else if (k == "syrk_small_tiles") g_opt.syrk_small_tiles = (int)v;
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

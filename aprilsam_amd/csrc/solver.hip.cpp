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
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
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
// ONE table of the options: name (aprilsam_amd_set_option / aprilsam_amd_get_option; the environment variable is APRILSAM_AMD_<NAME>), member,
// smallest accepted value, and whether the option is a host-side POLICY that no launch table or captured graph depends on (changing any
// other option bumps g_opt_epoch: every param re-plans and re-captures on its next call).
struct OptionDef { const char *name; int Options::*member; int min_value; bool policy; int max_value = 0x7fffffff; };
static const OptionDef OPTION_TABLE[] = {
    { "leaf_nodes", &Options::leaf_nodes, 1, false }, { "deterministic", &Options::deterministic, 0, true }, { "use_graph", &Options::use_graph, 0, true },
    { "device_timing", &Options::device_timing, 0, true }, { "trust_factor_cache", &Options::trust_factor_cache, 0, true },
    { "small_lds_kb", &Options::small_lds_kb, 0, false }, { "syrk_xcd_order", &Options::syrk_xcd_order, 0, false },
    { "schur_first", &Options::schur_first, 0, false }, { "syrk_small_tiles", &Options::syrk_small_tiles, 0, false }, { "syrk_pair_tiles", &Options::syrk_pair_tiles, 0, false }, { "syrk_group", &Options::syrk_group, 2, false, 8 }, { "panel_mode", &Options::panel_mode, 0, false },
    { "small_threads", &Options::small_threads, 64, false }, { "tp_fronts", &Options::tp_fronts, 0, false }, { "tp_lds_kb", &Options::tp_lds_kb, 0, false },
    { "tp_threads", &Options::tp_threads, 64, false }, { "inc_fast", &Options::inc_fast, 0, true }, { "inc_multi", &Options::inc_multi, 0, true },
    { "inc_one", &Options::inc_one, 0, true }, { "inc_one_up", &Options::inc_one_up, 1, true }, { "inc_one_dn", &Options::inc_one_dn, 1, true },
    { "inc_one_threads", &Options::inc_one_threads, 64, true }, { "inc_one_spin", &Options::inc_one_spin, 0, true }, { "inc_tail", &Options::inc_tail, 0, false },
    { "inc_inline", &Options::inc_inline, 0, true }, { "inc_update", &Options::inc_update, 0, true }, { "inc_tail_solve", &Options::inc_tail_solve, 0, true },
    { "inc_lazy_states", &Options::inc_lazy_states, 0, true }, { "inc_replan_tall", &Options::inc_replan_tall, 0, true },
    { "speculate_factors", &Options::speculate_factors, 0, true }, { "warm_up", &Options::warm_up, 0, true }, { "pin_last", &Options::pin_last, 0, false }, { "persist", &Options::persist, 0, false },
    { "persist_max_fronts", &Options::persist_max_fronts, 0, false }, { "linearize_staged_min", &Options::linearize_staged_min, 0, false },
    { "wave_backsolve", &Options::wave_backsolve, 0, false }, { "blk_backsolve", &Options::blk_backsolve, 0, false }, { "tail_poses", &Options::tail_poses, 8, false },
    { "batch_extend", &Options::batch_extend, 0, true }, { "extend_tail_fronts", &Options::extend_tail_fronts, 0, true }, { "mem_cap_mb", &Options::mem_cap_mb, 0, true },
    { "pool_guard", &Options::pool_guard, 0, false }, { "amalg", &Options::amalg, 0, false }, { "amalg_max", &Options::amalg_max, 1, false }, { "pool_poison", &Options::pool_poison, 0, false }, { "skip_flag_waits", &Options::skip_flag_waits, 0, false },
};
static const OptionDef *find_option(const char *name) {
    for (const OptionDef &d : OPTION_TABLE) if (!strcmp(d.name, name)) return &d;
    return nullptr;
}
static std::once_flag g_opt_once;
static void load_env_options() {
    std::call_once(g_opt_once, [] {
        for (const OptionDef &d : OPTION_TABLE) {
            std::string env = "APRILSAM_AMD_";
            for (const char *q = d.name; *q; q++) env += (char)toupper((unsigned char)*q);
            const char *sv = getenv(env.c_str());
            if (sv && *sv) g_opt.*(d.member) = std::min(d.max_value, std::max(d.min_value, (int)atof(sv)));
        }
    });
}

// ---- devices, slots and locks -------------------------------------------------------------------------------------------------
// A param (and the pack of the graph it is called with) is bound to a device SLOT: aprilsam_amd_param_set_device, default = the
// process default (aprilsam_amd_set_device / LOCAL_RANK / 0).  Slot s runs on HIP device s % device_count: on a node with N devices
// the slots 0 .. N-1 ARE the devices, and more slots than devices share devices.  Every entry point takes the lock of ITS slot
// (SlotLock) and makes that slot's device current on the calling thread: calls bound to different slots run concurrently -- one C
// process can drive N contexts on N devices from N threads -- calls bound to the same slot are serialised (as before: one lock).
// What is shared by all slots is guarded separately: the registries (param -> context, graph -> pack, param -> shard state: a
// lock per operation, node-based maps so that an element is not moved by another key's insertion), the planner's thread pool
// (symbolic.cpp: one plan at a time), the last-error record (errors.cpp).  The options (aprilsam_amd_set_option) are process-global
// and are meant to be set while no call is in flight.
constexpr int MAX_SLOTS = 64;
static std::atomic<int> g_device{ -1 };          // the process default slot (read by every call that is not bound to a slot, written by aprilsam_amd_set_device / the first call)
static std::mutex g_slot_mu[MAX_SLOTS];
static thread_local int t_slot = 0;               // slot of the call in progress on this thread
static int device_count() {
    static std::atomic<int> cached{ 0 };               // (a positive count does not change during the life of the process; zero is asked again)
    int n = cached.load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (n > 0) cached.store(n, std::memory_order_relaxed);
    return n;
}
static int physical_device(int slot) { const int n = device_count(); return n > 0 ? slot % n : 0; }
static thread_local int t_device = -1;             // the device this thread last made current through a SlotLock (hipSetDevice costs microseconds: 3 500 incremental steps feel it)
template <class V> struct Registry {             // std::map: iterators and elements stay put when OTHER keys come and go
    using Map = std::map<const void *, std::unique_ptr<V>>;
    Map m; mutable std::mutex mu;
    typename Map::iterator find(const void *k) { std::lock_guard<std::mutex> lk(mu); return m.find(k); }
    typename Map::iterator end() { return m.end(); }
    std::pair<typename Map::iterator, bool> emplace(const void *k, std::unique_ptr<V> v) { std::lock_guard<std::mutex> lk(mu); return m.emplace(k, std::move(v)); }
    void erase(typename Map::iterator it) { std::lock_guard<std::mutex> lk(mu); m.erase(it); }
    void put(const void *k, std::unique_ptr<V> v) { std::lock_guard<std::mutex> lk(mu); m[k] = std::move(v); }
    template <class F> void for_each(F &&f) { std::lock_guard<std::mutex> lk(mu); for (auto &kv : m) f(*kv.second); }
    template <class F> bool with(const void *k, F &&f) { std::lock_guard<std::mutex> lk(mu); auto it = m.find(k); if (it == m.end()) return false; f(*it->second); return true; }     // (the element cannot be erased while f looks at it)
};
static std::map<const void *, int> g_param_slot; static std::mutex g_param_slot_mu;      // aprilsam_amd_param_set_device
static std::atomic<int> g_param_slot_n{ 0 };      // bound params (zero: every call takes the default slot without looking)
static int default_slot() { const int d = g_device.load(std::memory_order_relaxed); return d < 0 ? 0 : d % MAX_SLOTS; }
static int slot_of_graph(const void *g);          // solver_pack.inc.h: the slot its pack lives on, or -1
static int slot_for(const void *param, const void *g) {
    if (param && g_param_slot_n.load(std::memory_order_acquire) > 0) { std::lock_guard<std::mutex> lk(g_param_slot_mu); auto it = g_param_slot.find(param); if (it != g_param_slot.end()) return it->second; }
    if (!param && g) { const int s = slot_of_graph(g); if (s >= 0) return s; }
    return default_slot();
}
struct SlotLock {
    int slot; std::unique_lock<std::mutex> lk;
    SlotLock(const void *param, const void *g) : slot(slot_for(param, g)), lk(g_slot_mu[slot]) {
        // (a call found by its graph alone -- april_graph_chi2, a graph being destroyed -- read the pack's slot before it held the lock: the
        // pack may have moved to another slot in between, pack_for below; look again under the lock)
        for (int now; (now = slot_for(param, g)) != slot; ) { lk.unlock(); slot = now; lk = std::unique_lock<std::mutex>(g_slot_mu[slot]); }
        t_slot = slot;
        if (device_count() > 0) {
            const int dev = physical_device(slot);
            int cur = -1;
            // (the caller may have changed this thread's device behind our back: ask, which is cheap; set only when it differs)
            if (t_device != dev || hipGetDevice(&cur) != hipSuccess || cur != dev) { (void)hipSetDevice(dev); t_device = dev; }
        }
    }
};
static std::once_flag g_dev_once;
static void ensure_device() {
    std::call_once(g_dev_once, [] {
        load_env_options();
        int n = device_count();
        if (n <= 0) fail(ERR_NO_DEVICE, "no HIP device visible: the april_graph_cholesky* / april_graph_chi2 entry points of "
                         "libaprilsam_amd.so run on an AMD GPU only (there is NO CPU fallback: nothing was computed)");
        if (g_device < 0) {
            const char *lr = getenv("LOCAL_RANK");
            g_device = lr ? atoi(lr) % n : 0;
        }
    });
}

// grow-only device / pinned-host buffers.  A failed allocation leaves the buffer empty (never dangling) and throws ERR_OOM;
// option mem_cap_mb (0 = off) refuses any single device buffer above that size the same way -- the tests use it to walk the
// out-of-memory path without exhausting a 288 GB device.
template <class T> struct DBuf {
    T *p = nullptr; size_t cap = 0;
    bool need(size_t n) {                                        // true: (re)allocated -- the contents are whatever the memory held before
        if (n <= cap) return false;
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
        return true;
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
// Runtime warm-up, once per process, from april_graph_cholesky_param_init (the API's set-up call; aprilsam.c:45-64 has nothing to set up).
// What the HIP runtime initialises lazily would otherwise land inside the first solver calls -- measured with rocprofv3 --hip-trace on the
// incremental demo: a stream 8-20 ms (each of the first two), the first host-to-device and the first device-to-host copy 7 ms each (the latter not before the first
// re-planned incremental step, 1 330 steps into the run), the first hipFuncSetAttribute (the code object) 2 ms, the first
// hipGraphInstantiate 8 ms.  Option warm_up = 0 leaves everything lazy.  Never fails: without a device, or on any error, it does nothing.
// ------------------------------------------------------------------------------------------------------
__global__ void k_warm_up(double *p) { if (threadIdx.x == 0) p[0] = 0.0; }        // (a name of its own in the profiles)
static std::once_flag g_warm_once;
void warm_up() noexcept {
    std::call_once(g_warm_once, [] {
        try {
            load_env_options();
            if (!g_opt.warm_up || device_count() <= 0) return;
            ensure_device();
            SlotLock lk(nullptr, nullptr);
            hipStream_t s = take_stream(t_slot), s2 = take_stream(t_slot);      // two: a graph owns its stream, and a program's second graph paid 10 ms for the second hardware queue inside its first solver call
            // copies of a size that takes the copy engines' path (small ones are done by a shader), in every direction the solver uses,
            // from pinned and from pageable host memory (the latter allocates the runtime's staging buffers)
            constexpr size_t WB = 4 << 20;
            char *d = nullptr, *h = nullptr;
            std::vector<char> pageable(WB, 0);
            if (hipMalloc((void **)&d, 2 * WB) == hipSuccess && hipHostMalloc((void **)&h, WB, hipHostMallocDefault) == hipSuccess) {
                memset(h, 0, WB);
                // (several sizes: the runtime copies small, medium and large blocks by different means -- inline, through staging buffers it
                // allocates on first use, by pinning the caller's pages -- and each of them has a first time)
                set_small_attr();
                auto kernel = [&] { hipLaunchKernelGGL(k_warm_up, dim3(1), dim3(64), 0, s, (double *)d + 8); };
                for (size_t nb : { (size_t)8 << 10, (size_t)32 << 10, (size_t)512 << 10, WB }) {
                    // ... each of them on an idle stream and behind a kernel that is still in flight (the runtime picks its means by that, too;
                    // the 7.6 ms of the demo's step 1 330 were a device-to-host copy's first time on one of these paths)
                    for (int busy = 0; busy < 2; busy++) {
                        auto before = [&] { if (busy) kernel(); else (void)hipStreamSynchronize(s); };
                        before(); (void)hipMemcpyAsync(d, h, nb, hipMemcpyHostToDevice, s);
                        before(); (void)hipMemcpyAsync(h, d, nb, hipMemcpyDeviceToHost, s);
                        before(); (void)hipMemcpyAsync(d + WB, d, nb, hipMemcpyDeviceToDevice, s);
                        before(); (void)hipMemcpyAsync(d, pageable.data(), nb, hipMemcpyHostToDevice, s);
                        before(); (void)hipMemcpyAsync(pageable.data(), d, nb, hipMemcpyDeviceToHost, s);
                    }
                }
                kernel(); (void)hipMemsetAsync(d, 0, WB, s);
                (void)hipStreamSynchronize(s);
                kernel();
                hipGraph_t graph = nullptr; hipGraphExec_t ge = nullptr;
                if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    kernel();
                    if (hipStreamEndCapture(s, &graph) == hipSuccess && graph) {
                        if (hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0) == hipSuccess && ge) (void)hipGraphLaunch(ge, s);
                        (void)hipGraphDestroy(graph);
                    }
                }
                (void)hipStreamSynchronize(s);
                if (ge) (void)hipGraphExecDestroy(ge);
            }
            if (d) (void)hipFree(d);
            if (h) (void)hipHostFree(h);
            if (s2) park_stream(t_slot, s2);
            park_stream(t_slot, s);
        } catch (...) {}
        (void)hipGetLastError();
    });
}

// ------------------------------------------------------------------------------------------------------
// Host-side consistency checks of the index arithmetic that host launch tables and device kernels share
// (kernels.hip.h: syrk_range / syrk_tiles / trapezoid decode, upd_packed_offset, block_tiles).  No GPU needed.
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
        auto run = [&](int s_lo, int s_hi, int tile, bool ahead = false) -> int {
            const SyrkRange g = syrk_range(R, C, ns, s_lo, s_hi, tile, ahead);
            const int nt = syrk_tiles(R, C, ns, s_lo, s_hi, tile, ahead);
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
        for (int tile : { TILE, TILE / 2 }) {          // both tile sizes of the wide update
            std::fill(applied.begin(), applied.end(), 0); std::fill(ksum.begin(), ksum.end(), 0);
            for (int o = 0; o * OBP < steps; o++) {
                // when outer block o is factored, each of its columns j must carry exactly the columns k < o * OBW from the wide updates
                // (the columns of its own block are applied inside k_block_chain / k_block_solve)
                const int k0 = o * OBP * NB, k1 = std::min(ns, k0 + OBP * NB);
                for (int j = k0; j < k1; j++)
                    for (int i = j; i < Rv; i++)
                        if (applied[(size_t)j * Rv + i] != k0 || ksum[(size_t)j * Rv + i] != (long long)k0 * (k0 - 1) / 2) return -13;
                const int rc = run(o * OBP, (o + 1) * OBP, tile); if (rc) return rc;
            }
            for (int j = ns; j < C; j++)
                for (int i = j; i < Rv; i++)
                    if (applied[(size_t)j * Rv + i] != ns || ksum[(size_t)j * Rv + i] != (long long)ns * (ns - 1) / 2) return -14;
        }
        for (int G : { 2, 3, 4 }) for (int tile : { TILE, TILE / 2 }) {          // ... and the same with GROUPS of outer blocks (options syrk_pair_tiles, syrk_group)
            std::fill(applied.begin(), applied.end(), 0); std::fill(ksum.begin(), ksum.end(), 0);
            const int nobs = (steps + OBP - 1) / OBP;
            for (int o = 0; o < nobs; o++) {
                const int k0 = o * OBP * NB, k1 = std::min(ns, k0 + OBP * NB);
                for (int j = k0; j < k1; j++)
                    for (int i = j; i < Rv; i++)
                        if (applied[(size_t)j * Rv + i] != k0 || ksum[(size_t)j * Rv + i] != (long long)k0 * (k0 - 1) / 2) return -15;
                int rc;
                const int g0 = (o - o % G) * OBP;
                if (o % G != G - 1) rc = run(g0, (o + 1) * OBP, tile, o + 1 < nobs);          // inside a group: the next block's columns only, K = the group so far (a front's last block: wide)
                else rc = run(g0, (o + 1) * OBP, tile);                                       // last of a group: everything to the right, K = the whole group
                if (rc) return rc;
            }
            for (int j = ns; j < C; j++)
                for (int i = j; i < Rv; i++)
                    if (applied[(size_t)j * Rv + i] != ns || ksum[(size_t)j * Rv + i] != (long long)ns * (ns - 1) / 2) return -16;
        }
        // (2) packed Schur update: offsets are the running count of (rows from the top of the diagonal block to the rhs row)
        long long run_off = 0;
        for (int j = ns; j < C; j++) { if (upd_packed_offset(R, ns, j) != run_off) return -21; run_off += Rv - 3 * (j / 3); }
        if (upd_packed_offset(R, ns, C) != run_off) return -22;
        // (3) row tiles of the row-solve kernel cover the rows below every outer block exactly once
        for (int o = 0; o * OBP < steps; o++)
            for (int rb : { 1, 2 }) {
                const int c1 = std::min(ns, (o + 1) * OBW), below = Rv - c1, nt = block_tiles(R, ns, o, rb);
                if ((long long)nt * BLOCK_ROWS * rb < below || (nt > 0 && (long long)(nt - 1) * BLOCK_ROWS * rb >= below)) return -31;
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
// bind a param (and, through it, the pack of the graph it is called with) to a device slot; whatever the param held is dropped
int api_param_set_device(const april_graph_cholesky_param_t *param, int slot) {
    if (!param || slot < 0 || slot >= MAX_SLOTS) return -1;
    { SlotLock lk0(param, nullptr); drop_shard_state(param); }      // (a sharded solve's buffers and communicator live on the old slot's device too)
    drop_context(param);                            // (under the lock of the slot it was on)
    std::lock_guard<std::mutex> lk(g_param_slot_mu);
    g_param_slot[param] = slot;
    g_param_slot_n.store((int)g_param_slot.size(), std::memory_order_release);
    return 0;
}
int api_param_get_device(const april_graph_cholesky_param_t *param) { return physical_device(slot_for(param, nullptr)); }
// a param that is re-initialised or destroyed loses its binding (another param allocated at the same address later must not inherit it)
void unbind_param(const april_graph_cholesky_param_t *param) {
    if (g_param_slot_n.load(std::memory_order_acquire) == 0) return;
    std::lock_guard<std::mutex> lk(g_param_slot_mu);
    g_param_slot.erase(param);
    g_param_slot_n.store((int)g_param_slot.size(), std::memory_order_release);
}
int api_set_option(const char *name, double v) {
    load_env_options();
    const OptionDef *d = name ? find_option(name) : nullptr;
    if (!d) return -1;
    const int nv = std::min(d->max_value, std::max(d->min_value, (int)v));
    if (g_opt.*(d->member) == nv) return 0;
    g_opt.*(d->member) = nv;
    if (!d->policy) g_opt_epoch++;
    return 0;
}
int api_get_option(const char *name, double *v) {
    load_env_options();
    const OptionDef *d = name ? find_option(name) : nullptr;
    if (!d || !v) return -1;
    *v = (double)(g_opt.*(d->member));
    return 0;
}

}  // namespace asam

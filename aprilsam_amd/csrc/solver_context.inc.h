// solver_context.inc.h -- part of solver.hip.cpp (ONE translation unit: the kernels of kernels.hip.h are compiled once); included from there,
// inside namespace asam.  Contents: the solver context of a param (plan, device tables, fronts, captured graphs), the failure path, plan upload, launch tables and the numeric phase of a batch step.
// ------------------------------------------------------------------------------------------------------
// solver context — one per april_graph_cholesky_param_t pointer
// ------------------------------------------------------------------------------------------------------
enum { K_LINEARIZE = 0, K_FRONT_SMALL, K_ASSEMBLE_BIG, K_PANEL_BIG, K_SYRK_BIG, K_BACKSOLVE, K_UPDATE, NKERN };      // K_PANEL_BIG = k_block_chain + k_block_solve, K_SYRK_BIG = k_syrk_big + k_syrk_big32, K_BACKSOLVE = every back-substitution kernel
static const char *const KNAMES[NKERN] = { "k_linearize", "k_front_small", "k_assemble_big", "k_panel_big", "k_syrk_big", "k_backsolve", "k_update_states" };
struct Launch { int list_off, pre_off, n, grid; bool single = false; int tile = TILE; };   // single: every front has exactly one work item     // offsets into the int launch-table buffer

struct LevelPlan {
    int small_off = 0, n_small = 0; size_t small_lds = 0;      // fronts handled by k_front_small
    long long full_limit = 0;                                  // ... fully in LDS when their array fits this many bytes, else panel mode
    int small_nt = 512;                                        // ... with this many threads per workgroup
    int n_big = 0; size_t asm_lds = 0;
    Launch asm_big{};                                          // k_assemble_big (chunks of block columns)
    std::vector<Launch> syrka, syrk1;                          // paired outer blocks (option syrk_pair_tiles; empty: every block closes with syrkw): after the FIRST block of a pair the "ahead" update of the fronts that have a second one, and the single-block wide update of the fronts that end here; syrkw then holds the K = 256 update after the second block
    bool paired = false;
    std::vector<Launch> bchain, btile, syrkw;                  // per 128-column outer block: diagonal-block workgroups, row tiles (Launch::tile = rows per wave / 16), the wide update that closes it (Launch::tile = output tile)
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
    bool pristine = false;                  // ... and no incremental / extended-plan step has touched them since (inc_prepare need not run again for the same plan)
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
    std::vector<hipEvent_t> k_ev; std::vector<int> k_ids, k_lev;
    std::vector<double> lev_up_ms, lev_dn_ms;       // ... the same event pairs summed per level of the assembly tree (factorisation / back substitution)
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
    long long pat_serial = -1, pat_topo = -1;      // the pack (serial) and its topo_version at which `pat` last equalled the packed endpoints (prepare_plan)
    RefModel model;                       // the reference's tree / counters (refmodel.cpp), rebuilt lazily after a batch
    PatchList patches;                    // per-step table updates of the incremental fast path
    int batch_factors = 0;                // #factors at the last batch step
    // Factors with an information matrix that is not symmetric as given (GraphPack::asym): wt[f] = 1 when the reference eliminates endpoint b
    // before endpoint a, i.e. when the block it accumulates is J_b^T W J_a (aprilsam.c:171,520: upper triangle of ITS order only) and not
    // J_a^T W J_b -- the two are transposes of each other only for a symmetric W.  Fixed when the factor enters the system: at a batch call
    // for every factor (the reference re-orders everything), at an incremental call for the new ones (old poses keep their positions, new
    // ones are appended, aprilsam.c:393-396).  Travels to the device as bit 1 of the per-factor swap byte (k_linearize).
    std::vector<unsigned char> wt; bool wt_any = false, wt_dirty = false;
    long long wt_serial = -1, wt_topo = -1, wt_content = -1;
    DBuf<long long> d_guard; DBuf<int> d_guard_cnt; int n_guard = 0, guard_len = 0;      // option pool_guard: offsets of the guard bands in d_pool
    std::vector<unsigned char> swap_host;          // host copy of d_swap's base-plan part (the source of an asynchronous copy: must outlive it)
    // captured numeric phase
    // multi-level ("persistent") launches of the batch path: the top levels of the tree, where a level holds only a handful
    // of fronts, run as ONE launch for the factorisation and ONE for the back substitution, fronts waiting on per-front
    // dependency flags instead of on kernel boundaries (kernels.hip.h: wait_flag / publish_flag)
    int persist_l0 = -1;                  // first level of the multi-level launch, -1: none
    int p_up_off = 0, p_up_n = 0, p_dn_off = 0, p_dn_n = 0, p_nt = 1024; size_t p_up_lds = 0, p_dn_lds = 0; long long p_up_full = 0; int p_dn_maxns = 0;
    DBuf<int> d_flags, d_flevel, d_perm, d_epoch, d_marks;   // d_epoch: the step counter every dependency flag carries (kernels.hip.h wait_flag); d_marks: fronts regenerated by an incremental step
    int flag_stride = 0;                  // d_flags = three arrays of this many words: "factor done", "x done", "vectors ready" per front
    long long epoch_steps = 0;            // numeric phases enqueued since the counter was (re)started (rewind_epoch)
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
    int api_key_runs = 0;                          // calls seen with this key: the first one runs without a graph (below)
    // captured graphs that are no longer current: hipGraphExecDestroy takes 0.24 ms on this stack, so they are destroyed while the
    // GPU works on a step (reap_retired), not on the way to the next plan
    // A retired graph may still have launches in flight (resident / sharded loops enqueue steps without a sync in between): it carries an
    // event recorded on the stream it was last launched on and is destroyed only once that event has completed.
    struct Retired { hipGraphExec_t g; hipEvent_t done; };
    std::vector<Retired> retired;
    hipStream_t graph_stream = nullptr;            // the stream gexec / gexec_api were last launched on (run_numeric)
    void reap_retired(bool wait = false) {
        size_t keep = 0;
        for (Retired &r : retired) {
            if (r.done) {
                if (wait) (void)hipEventSynchronize(r.done);
                else if (hipEventQuery(r.done) != hipSuccess) { retired[keep++] = r; continue; }
                (void)hipEventDestroy(r.done);
            }
            (void)hipGraphExecDestroy(r.g);
        }
        retired.resize(keep);
    }
    void retire(hipGraphExec_t &g) {
        if (!g) return;
        if (retired.size() >= 8) reap_retired();                   // (bounded: call sequences that never reach a reaping point)
        if (retired.size() >= 32) reap_retired(true);              // (... and whose launches never finish in between)
        Retired r{ g, nullptr };
        if (graph_stream && hipEventCreateWithFlags(&r.done, hipEventDisableTiming) == hipSuccess) {
            if (hipEventRecord(r.done, graph_stream) != hipSuccess) { (void)hipEventDestroy(r.done); r.done = nullptr; }
        } else r.done = nullptr;
        retired.push_back(r); g = nullptr;
    }
    double lambda_val = -1; int lambda_N = -1;     // what d_lambda currently holds (uniform batch value), -1: unknown
    void release() {
        d_i32.release(); d_fd.release(); d_dest.release(); d_child.release(); d_lambda.release(); d_tab.release(); d_swap.release(); d_pos.release();
        d_pool.release(); d_H.release(); d_x.release(); d_diag.release(); d_bad.release(); h_bad.release(); patches.release();
        h_done.release(); h_kstamp.release(); d_prof.release(); d_upd.release(); d_wbuf.release(); d_flags.release(); d_flevel.release(); d_epoch.release(); d_marks.release(); d_perm.release(); d_solve_tab.release(); d_dinv.release(); d_bsb_far.release(); d_bsb_flags.release(); d_guard.release(); d_guard_cnt.release(); n_guard = 0;
        retire(gexec); retire(gexec_api); reap_retired(true);
        if (have_events) for (auto &e : ev) (void)hipEventDestroy(e);
        have_events = false;
        for (auto &e : k_ev) (void)hipEventDestroy(e);
        k_ev.clear();
    }
};
static Registry<Context> g_ctx;

static void forget_stream(hipStream_t s) {
    g_ctx.for_each([&](Context &c) { if (c.graph_stream == s) c.graph_stream = nullptr; });
}

static Context &ctx_for(const april_graph_cholesky_param_t *p) {
    auto it = g_ctx.find(p);
    if (it == g_ctx.end()) it = g_ctx.emplace(p, std::make_unique<Context>()).first;
    return *it->second;
}
void drop_context(const april_graph_cholesky_param_t *p) {
    SlotLock lk(p, nullptr);
    auto it = g_ctx.find(p);
    if (it != g_ctx.end()) { it->second->release(); g_ctx.erase(it); }
}
bool get_stats(const april_graph_cholesky_param_t *p, aprilsam_amd_stats_t *out) {
    SlotLock lk(p, nullptr);
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
    SlotLock lk(param, g);
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
        g_ctx.put(param, std::move(fresh));
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

// The step counter behind the dependency flags (kernels.hip.h wait_flag) starts here and is advanced once per numeric phase by the phase's
// first kernel.  It is a 32-bit word: a context that has enqueued 2^30 phases (days of back-to-back iterations) starts it over, behind a
// stream synchronisation and with every flag and mark zeroed again (rewind_epoch, called where phases are enqueued).
static const int FIRST_EPOCH = 1 << 20;
struct Context;
static void rewind_epoch(Context &c, hipStream_t s, long long phases);
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

// doubles of d_diag a level needs: four NB x NB inverse blocks per active front (the fronts that keep none of their own, FrontDesc::dinv0)
static size_t diag_doubles(int n_big) { return (size_t)std::max(n_big, 1) * OBP * NB * NB; }
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
        if (!wide.empty() && wgs <= BSB_MAX_WGS && g_opt.blk_backsolve) {
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
    const int steps = (3 * nsb_of(big[0]) + NB - 1) / NB;
    auto active = [&](int sidx) { int nact = 0; while (nact < (int)big.size() && 3 * nsb_of(big[nact]) > sidx * NB) nact++; return nact; };
    // the wide update that closes every outer block of OBP panels (K = the block's columns, everything to its right) -- or, on levels
    // whose wide updates are large (option syrk_pair_tiles), every PAIR of outer blocks (syrk_range)
    const int nobs = (steps + OBP - 1) / OBP;
    auto make_sub = [&](int i0, int i1, auto count) {          // like make() for the fronts big[i0 .. i1)
        Launch La; La.list_off = list_off + i0; La.n = i1 - i0; La.pre_off = (int)tab.size();
        int acc = 0; tab.push_back(0);
        for (int i = i0; i < i1; i++) { acc += count(big[i]); tab.push_back(acc); }
        La.grid = acc; La.single = acc == La.n;
        return La;
    };
    auto tile_for = [&](int i0, int i1, int s_lo, int s_hi, bool ahead) {
        int tile = TILE;
        if (g_opt.syrk_small_tiles > 0) {         // few tiles: 32 x 32 ones (k_syrk_big32)
            long long nt64 = 0;
            for (int i = i0; i < i1; i++) nt64 += syrk_tiles(rows(big[i]), cols(big[i]), 3 * nsb_of(big[i]), s_lo, s_hi, TILE, ahead);
            if (nt64 < g_opt.syrk_small_tiles) tile = TILE / 2;
        }
        return tile;
    };
    {
        long long nt0 = 0;
        for (int i = 0; i < active(0); i++) nt0 += syrk_tiles(rows(big[i]), cols(big[i]), 3 * nsb_of(big[i]), 0, OBP, TILE);
        L.paired = g_opt.syrk_pair_tiles > 0 && nobs >= 2 && nt0 >= g_opt.syrk_pair_tiles;
    }
    for (int o = 0; o < nobs; o++) {
        const int s_lo = o * OBP, s_hi = std::min(s_lo + OBP, steps), nact = active(s_lo);
        if (!L.paired) {
            const int tile = tile_for(0, nact, s_lo, s_hi, false);
            L.syrkw.push_back(make(nact, [&](int t) { return syrk_tiles(rows(t), cols(t), 3 * nsb_of(t), s_lo, s_hi, tile); }));
            L.syrkw.back().tile = tile;
            continue;
        }
        const int G = std::max(2, g_opt.syrk_group), p_lo = (o - o % G) * OBP;       // K = the blocks of the group so far
        if (o % G != G - 1) {
            // not the last block of its group: fronts with a next block (a prefix of `big`, sorted by own columns) update that block's
            // columns only; fronts whose last block this is get their wide update now
            const int n2 = o + 1 < nobs ? active((o + 1) * OBP) : 0;
            int tile = tile_for(0, n2, p_lo, s_hi, true);
            L.syrka.push_back(make_sub(0, n2, [&](int t) { return syrk_tiles(rows(t), cols(t), 3 * nsb_of(t), p_lo, s_hi, tile, true); })); L.syrka.back().tile = tile;
            tile = tile_for(n2, nact, p_lo, s_hi, false);
            L.syrk1.push_back(make_sub(n2, nact, [&](int t) { return syrk_tiles(rows(t), cols(t), 3 * nsb_of(t), p_lo, s_hi, tile); })); L.syrk1.back().tile = tile;
            L.syrkw.push_back(Launch{ 0, 0, 0, 0, false });
        } else {
            const int tile = tile_for(0, nact, p_lo, s_hi, false);
            L.syrka.push_back(Launch{ 0, 0, 0, 0, false }); L.syrk1.push_back(Launch{ 0, 0, 0, 0, false });
            L.syrkw.push_back(make(nact, [&](int t) { return syrk_tiles(rows(t), cols(t), 3 * nsb_of(t), p_lo, s_hi, tile); }));
            L.syrkw.back().tile = tile;
        }
    }
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
// per-factor swap byte of the base plan: bit 0 = the off-diagonal block is stored transposed (Plan::fac_swap), bit 1 = Context::wt
static void fill_swap_host(Context &c) {
    const Plan &P = c.plan;
    c.swap_host.assign(P.fac_swap.begin(), P.fac_swap.end());
    if (c.wt_any) for (size_t f = 0; f < c.swap_host.size() && f < c.wt.size(); f++) c.swap_host[f] |= (unsigned char)(c.wt[f] << 1);
}
static void upload_plan(Context &c, hipStream_t s, const ShardLayout *lay = nullptr) {
    const Plan &P = c.plan;
    const bool uprof = getenv("APRILSAM_AMD_PLAN_PROFILE") != nullptr;
    const double u0 = uprof ? now_ms() : 0;
    c.retire(c.gexec); c.retire(c.gexec_api); c.api_key_runs = 0;
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
    for (int t = 0; t < P.nF; t++) {          // the child with the largest update block (k_assemble_big stores it instead of adding it to zeros)
        int best = -1, second = -1;
        for (int k = P.ch_ptr[t]; k < P.ch_ptr[t + 1]; k++) if (P.f_nub[P.ch_idx[k]] > 0 && (best < 0 || P.f_nub[P.ch_idx[k]] > P.f_nub[P.ch_idx[best]])) best = k;
        for (int k = P.ch_ptr[t]; k < P.ch_ptr[t + 1]; k++) if (k != best && P.f_nub[P.ch_idx[k]] > 0 && (second < 0 || P.f_nub[P.ch_idx[k]] > P.f_nub[P.ch_idx[second]])) second = k;
        fd[t].prim1 = best < 0 ? 0 : best - P.ch_ptr[t] + 1; fd[t].prim2 = second < 0 ? 0 : second - P.ch_ptr[t] + 1;
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
    const double u1 = uprof ? now_ms() : 0;
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
    d.schur_first_nub = std::max(0, g_opt.schur_first);
    d.epoch = nullptr; d.flevel = nullptr; d.l0 = 0; d.marks = nullptr;      // (epoch: below, once the counter exists; flevel / l0 / marks: set in the copies the multi-level launches take)
    if (getenv("APRILSAM_AMD_KPROF")) { c.d_prof.need((size_t)PROF_SLOTS * P.nF); HIPCHECK(hipMemsetAsync(c.d_prof.p, 0, (size_t)8 * PROF_SLOTS * P.nF, s)); d.prof = c.d_prof.p; d.prof_mode = atoi(getenv("APRILSAM_AMD_KPROF")) >= 2 ? atoi(getenv("APRILSAM_AMD_KPROF")) : 1; }
    c.d_swap.need((size_t)P.F + INC_FACT_); c.d_pos.need((size_t)P.N + INC_NODES_);
    fill_swap_host(c);
    HIPCHECK(hipMemcpyAsync(c.d_swap.p, c.swap_host.data(), P.F, hipMemcpyHostToDevice, s));
    c.wt_dirty = false;
    HIPCHECK(hipMemcpyAsync(c.d_pos.p, P.pos.data(), (size_t)P.N * 4, hipMemcpyHostToDevice, s));
    c.d_perm.need((size_t)P.N + INC_NODES_);
    HIPCHECK(hipMemcpyAsync(c.d_perm.p, P.perm.data(), (size_t)P.N * 4, hipMemcpyHostToDevice, s));

    const double u2 = uprof ? now_ms() : 0;
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
    {
        // No flag is ever reset: a finished front publishes the step number, which only grows (a flag that is read late or stale can then only
        // mean "not yet").  New flag memory must therefore start BELOW every step number: zeroed whenever the buffer was (re)allocated --
        // judged by DBuf::need itself, not by the pointer (hipFree + hipMalloc may hand the same address back for the larger block: round-5
        // advisor finding) -- and the counter starts once per context, far above zero.
        const size_t stride = (size_t)P.nF + MAX_TAIL_FRONTS;
        const bool fresh = c.d_flags.need(3 * stride);            // done / x done / vectors ready
        c.d_flevel.need(stride);
        if (fresh || (size_t)c.flag_stride != stride) HIPCHECK(hipMemsetAsync(c.d_flags.p, 0, c.d_flags.cap * 4, s));
        c.flag_stride = (int)stride;
        if (c.d_marks.need(stride)) HIPCHECK(hipMemsetAsync(c.d_marks.p, 0, c.d_marks.cap * 4, s));
        if (c.d_epoch.need(1)) { HIPCHECK(hipMemcpyAsync(c.d_epoch.p, &FIRST_EPOCH, 4, hipMemcpyHostToDevice, s)); c.epoch_steps = 0; }
        d.epoch = c.d_epoch.p;
    }
    if (inc) { c.d_upd.need((size_t)g_opt.persist_max_fronts + 64); c.d_wbuf.need((size_t)1 << 19); }
    HIPCHECK(hipMemcpyAsync(c.d_flevel.p, P.f_level.data(), (size_t)P.nF * 4, hipMemcpyHostToDevice, s));
    if (tab.empty()) tab.push_back(0);
    c.d_tab.need(tab.size() + INC_TAB_);
    c.inc.tab_used = (long long)tab.size();
    c.base_tab = tab;
    const double u3 = uprof ? now_ms() : 0;
    HIPCHECK(hipMemcpyAsync(c.d_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHECK(hipStreamSynchronize(s));      // host vectors above go out of scope
    const double u4 = uprof ? now_ms() : 0;

    const long long pool_slack = inc ? std::max<long long>(INC_POOL_MIN, P.pool_doubles / 4) : 0;
    const long long pool_doubles = lay ? lay->pool_doubles : P.pool_doubles;
    c.d_pool.need((size_t)std::max<long long>(pool_doubles, 1) + (size_t)pool_slack);
    c.inc.pool_used = pool_doubles; c.inc.pool_cap = (long long)c.d_pool.cap;
    c.n_guard = 0;
    if (g_opt.pool_guard > 0 && !lay) {
        const long long G = ((long long)g_opt.pool_guard + 31) & ~31ll;
        std::vector<long long> go((size_t)P.nF);
        for (int t = 0; t < P.nF; t++) go[t] = P.f_off[t] + ((((long long)P.rows(t) * P.cols(t)) + 31) & ~31ll);
        c.d_guard.need((size_t)P.nF); c.d_guard_cnt.need(2);
        HIPCHECK(hipMemcpyAsync(c.d_guard.p, go.data(), go.size() * 8, hipMemcpyHostToDevice, s));
        HIPCHECK(hipMemsetAsync(c.d_guard_cnt.p, 0, 8, s));
        c.n_guard = P.nF; c.guard_len = (int)G;
        hipLaunchKernelGGL(k_guard, dim3(c.n_guard), dim3(TPB), 0, s, 0, c.d_guard.p, c.n_guard, c.guard_len, c.d_pool.p, c.d_guard_cnt.p);
        HIPCHECK(hipStreamSynchronize(s));      // (go goes out of scope)
    }
    c.d_H.need((size_t)9 * ((size_t)std::max(1, P.n_slots) + (size_t)5 * INC_FACT_)); c.d_x.need((size_t)3 * ((size_t)P.N + INC_NODES_ + 1));
    c.inc.zpos = P.N + INC_NODES_;
    HIPCHECK(hipMemsetAsync(c.d_x.p + (size_t)3 * c.inc.zpos, 0, 24, s));
    c.inc.slots_used = P.n_slots;
    c.inc.ready = false; c.inc.t_first.clear(); c.same_topo_batches = 0;
    c.d_bad.need(4); c.h_bad.need(4);
    {   // (a param that is used incrementally: fronts near the root collect the rows of every loop closure since the plan was made and
        // may outgrow the single-workgroup kernel -- room for a few of them on the multi-workgroup path, whose scratch a plan without
        // such fronts would not have; measured on the M3500 demo: 13 steps re-planned for 70 KB of scratch)
        size_t mx = inc ? diag_doubles(160) : 1;          // (incremental steps: room for the big fronts of a regenerated level; a step that needs more re-plans, inc_fail(11))
        for (int l = 0; l < P.nLevels; l++) mx = std::max(mx, diag_doubles(c.levels[l].n_big));
        c.d_diag.need(mx);
    }
    if (uprof) fprintf(stderr, "aprilsam_amd upload: graphs destroyed + descriptors %.3f, index arrays + copies %.3f, launch tables %.3f, last copies + sync %.3f, pools %.3f ms\n", u1 - u0, u2 - u1, u3 - u2, u4 - u3, now_ms() - u4);
    c.st.n_fronts = P.nF; c.st.n_levels = P.nLevels; c.st.max_front_rows = P.max_rows;
    c.st.nnz_L = P.nnzL; c.st.flops_factor = P.flops; c.st.bytes_fronts = 8.0 * (double)pool_doubles;
}

static void set_small_attr() {
    static std::once_flag once[MAX_SLOTS];          // (function attributes are kept per device)
    std::call_once(once[physical_device(t_slot) % MAX_SLOTS], [] {
        HIPCHECK(hipFuncSetAttribute((const void *)k_front_small<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_front_small<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_front_small<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_backsolve_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_backsolve_t<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_backsolve_w, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_backsolve_t<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_block_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHECK(hipFuncSetAttribute((const void *)k_backsolve_blk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
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
// the plan as the batch path's two multi-level launches see it: flags carry the iteration number
static DevPlan persist_plan(const Context &c) { DevPlan d = c.dp; d.flevel = c.d_flevel.p; d.l0 = c.persist_l0; return d; }
static void rewind_epoch(Context &c, hipStream_t s, long long phases) {
    c.epoch_steps += phases;
    if (c.epoch_steps < (1ll << 30) || !c.d_epoch.p) return;
    HIPCHECK(hipStreamSynchronize(s));                      // nothing in flight reads a flag
    HIPCHECK(hipMemsetAsync(c.d_flags.p, 0, c.d_flags.cap * 4, s));
    HIPCHECK(hipMemsetAsync(c.d_marks.p, 0, c.d_marks.cap * 4, s));
    HIPCHECK(hipMemcpyAsync(c.d_epoch.p, &FIRST_EPOCH, 4, hipMemcpyHostToDevice, s));
    c.epoch_steps = phases;
}
// debug option pool_poison (kernels.hip.h k_poison): NaN into everything the step's launches hand from one workgroup to another
static void enqueue_poison(Context &c, hipStream_t s, const int *list, int n, int what = 3, const UpdRec *recs = nullptr) {
    if (g_opt.pool_poison <= 0 || n <= 0) return;
    static_assert(offsetof(UpdRec, mode) % 4 == 0 && sizeof(UpdRec) % 4 == 0, "UpdRec::mode as a strided int");
    hipLaunchKernelGGL(k_poison, dim3(n), dim3(TPB), 0, s, c.dp, list, n, recs ? (const int *)((const char *)recs + offsetof(UpdRec, mode)) : (const int *)nullptr,
                       (int)(sizeof(UpdRec) / 4), what, c.d_pool.p, c.d_x.p);
}
static void launch_front_persist(Context &c, hipStream_t s) {
    const int *list = c.d_tab.p + c.p_up_off;
    int *fl = c.d_flags.p;
    DevPlan dpe = persist_plan(c);
    if (g_opt.skip_flag_waits > 0) dpe.l0 = 1 << 30;       // debug (negative control of pool_poison): no front of this launch waits for its children
    if (c.p_nt >= 1024) hipLaunchKernelGGL(k_front_small<1024>, dim3(c.p_up_n), dim3(1024), c.p_up_lds, s, dpe, list, c.d_pool.p, c.d_H.p, c.d_bad.p, c.p_up_full, fl, 1);
    else if (c.p_nt >= 512) hipLaunchKernelGGL(k_front_small<512>, dim3(c.p_up_n), dim3(512), c.p_up_lds, s, dpe, list, c.d_pool.p, c.d_H.p, c.d_bad.p, c.p_up_full, fl, 1);
    else hipLaunchKernelGGL(k_front_small<256>, dim3(c.p_up_n), dim3(256), c.p_up_lds, s, dpe, list, c.d_pool.p, c.d_H.p, c.d_bad.p, c.p_up_full, fl, 1);
}

static void launch_front_small(Context &c, const LevelPlan &L, hipStream_t s, const int *tab = nullptr) {
    if (!tab) tab = c.d_tab.p;
    const int nt = L.small_nt;
    if (nt >= 1024) hipLaunchKernelGGL(k_front_small<1024>, dim3(L.n_small), dim3(1024), L.small_lds, s, c.dp, tab + L.small_off, c.d_pool.p, c.d_H.p, c.d_bad.p, L.full_limit, (int *)nullptr, 0);
    else if (nt >= 512) hipLaunchKernelGGL(k_front_small<512>, dim3(L.n_small), dim3(512), L.small_lds, s, c.dp, tab + L.small_off, c.d_pool.p, c.d_H.p, c.d_bad.p, L.full_limit, (int *)nullptr, 0);
    else hipLaunchKernelGGL(k_front_small<256>, dim3(L.n_small), dim3(256), L.small_lds, s, c.dp, tab + L.small_off, c.d_pool.p, c.d_H.p, c.d_bad.p, L.full_limit, (int *)nullptr, 0);
}

// The big fronts of one level, 128 columns (an outer block of OBP panels) at a time: diagonal block in LDS with the inverses of its four
// 32 x 32 diagonal blocks as a by-product (k_block_chain), row solves on the matrix cores (k_block_solve), then ONE wide update of
// everything to the right with K = the block's columns (k_syrk_big / k_syrk_big32).
template <class Tic, class Toc>
static void enqueue_big_steps(Context &c, const LevelPlan &L, hipStream_t s, Tic tic, Toc toc, const int *tab = nullptr) {
    if (!tab) tab = c.d_tab.p;
    const int xcd = std::max(0, g_opt.syrk_xcd_order) << SYRK_MODE_XCD_SHIFT;
    for (size_t o = 0; o < L.bchain.size(); o++) {
        const Launch &bc = L.bchain[o], &bt = L.btile[o], &sw = L.syrkw[o];
        tic(K_PANEL_BIG);
        hipLaunchKernelGGL(k_block_chain, dim3(bc.n), dim3(BCH_THREADS), block_chain_lds(), s, c.dp, tab + bc.list_off, (int)o, c.d_pool.p, c.d_diag.p, c.d_dinv.p, c.d_bad.p);
        if (bt.grid > 0) {
            if (bt.tile == 2) hipLaunchKernelGGL(k_block_solve<2>, dim3(bt.grid), dim3(TPB), block_solve_lds(), s, c.dp, tab + bt.list_off, tab + bt.pre_off, bt.n, (int)o, c.d_pool.p, c.d_diag.p, c.d_dinv.p);
            else hipLaunchKernelGGL(k_block_solve<1>, dim3(bt.grid), dim3(TPB), block_solve_lds(), s, c.dp, tab + bt.list_off, tab + bt.pre_off, bt.n, (int)o, c.d_pool.p, c.d_diag.p, c.d_dinv.p);
        }
        toc();
        auto wide = [&](const Launch &w, int s_lo, int s_hi, int mode) {
            if (w.grid <= 0) return;
            tic(K_SYRK_BIG);
            // (s_lo, s_hi) in panel steps: the kernel clips s_hi * NB to the front's own columns
            if (w.tile == TILE / 2) hipLaunchKernelGGL(k_syrk_big32, dim3(w.grid), dim3(TPB), 0, s, c.dp, tab + w.list_off, tab + w.pre_off, w.n, s_lo, s_hi, mode | xcd, c.d_pool.p);
            else hipLaunchKernelGGL(k_syrk_big, dim3(w.grid), dim3(TPB), 0, s, c.dp, tab + w.list_off, tab + w.pre_off, w.n, s_lo, s_hi, mode | xcd, c.d_pool.p);
            toc();
        };
        const int G = std::max(2, g_opt.syrk_group), g0 = (int)(o - o % G) * OBP;
        if (!L.paired) wide(sw, (int)o * OBP, (int)(o + 1) * OBP, 1);
        else if ((int)(o % G) != G - 1) { wide(L.syrka[o], g0, (int)(o + 1) * OBP, 2); wide(L.syrk1[o], g0, (int)(o + 1) * OBP, 1); }
        else wide(sw, g0, (int)(o + 1) * OBP, 1);
    }
}

// kernels of one level of the factorisation (small LDS fronts, big multi-workgroup path)
template <class Tic, class Toc>
static void enqueue_factor_level(Context &c, const LevelPlan &L, hipStream_t s, Tic tic, Toc toc, const int *tab = nullptr) {
    if (!tab) tab = c.d_tab.p;
    if (L.n_small) {
        tic(K_FRONT_SMALL);
        launch_front_small(c, L, s, tab);
        toc();
    }
    if (L.n_big) {
        tic(K_ASSEMBLE_BIG);
        hipLaunchKernelGGL(k_assemble_big, dim3(L.asm_big.grid), dim3(TPB), L.asm_lds, s, c.dp, tab + L.asm_big.list_off,
                           tab + L.asm_big.pre_off, L.asm_big.n, c.d_pool.p, c.d_H.p);
        toc();
        enqueue_big_steps(c, L, s, tic, toc, tab);
    }
}

// enqueue: linearise -> per level {assemble+factor} -> back substitution -> state update
// ev != null: record stage events (0 start, 1 after linearise, 2 after factor, 3 after solve+update)
// ktime: bracket EVERY kernel launch with its own HIP event pair on this stream (c.k_ev / c.k_ids)
static void enqueue_numeric(Context &c, GraphPack &gp, hipStream_t s, hipEvent_t *ev, bool unary_at_lp = false, bool ktime = false, bool io_host = false, bool relin = false) {
    const Plan &P = c.plan;
    const int F = P.F, N = P.N;
    size_t nev = 0;
    if (ktime) { c.k_ids.clear(); c.k_lev.clear(); }
    int cur_level = -1;                              // (profile: the level the launches that follow belong to; multi-level launches: their first level)
    auto tic = [&](int id) {
        if (!ktime) return;
        if (c.k_ev.size() < nev + 2) { c.k_ev.resize(nev + 2); HIPCHECK(hipEventCreate(&c.k_ev[nev])); HIPCHECK(hipEventCreate(&c.k_ev[nev + 1])); }
        HIPCHECK(hipEventRecord(c.k_ev[nev], s));
        c.k_ids.push_back(id); c.k_lev.push_back(cur_level);
    };
    auto toc = [&]() { if (ktime) { HIPCHECK(hipEventRecord(c.k_ev[nev + 1], s)); nev += 2; } };
    if (ev) HIPCHECK(hipEventRecord(ev[0], s));
    if (c.dp.prof) HIPCHECK(hipMemsetAsync(c.d_prof.p, 0, (size_t)8 * PROF_SLOTS * P.nF, s));
    if (io_host) hipLaunchKernelGGL(k_load_states, dim3((3 * N + TPB - 1) / TPB), dim3(TPB), 0, s, 3 * N, gp.h_state.p, gp.d_state.p, gp.d_lp.p);
    enqueue_poison(c, s, nullptr, P.nF);
    tic(K_LINEARIZE);
    {   // (a variant of the kernel without the asymmetric-W orientation branch, for graphs that have no such factor, was measured in round 6: no
        // difference -- 0.79 ms on the 1 M lattice either way)
        auto launch = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((F + TPB - 1) / TPB), dim3(TPB), 0, s, 0, F, (const int *)nullptr, gp.d_fa.p, gp.d_fb.p, gp.d_z.p, gp.d_W.p,
                               gp.d_lp.p, gp.d_state.p, c.d_swap.p, c.dp.slot_blk, c.dp.slot_rhs, c.d_H.p, c.d_bad.p, unary_at_lp ? gp.d_upt.p : (const double *)nullptr, c.d_epoch.p);
        };
        if (F >= g_opt.linearize_staged_min) launch(k_linearize_t<true>); else launch(k_linearize_t<false>);
    }
    if (!gp.host_idx.empty()) {         // host-evaluated factors: their blocks replace the null contributions written above
        const int nh = (int)gp.host_idx.size();
        HIPCHECK(hipMemcpyAsync(gp.d_hostH.p, gp.h_hostH.p, (size_t)33 * 8 * nh, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_scatter_host, dim3((nh + TPB - 1) / TPB), dim3(TPB), 0, s, nh, gp.d_host_idx.p, gp.d_hostH.p, gp.d_fb.p, c.d_swap.p,
                           c.dp.slot_blk, c.dp.slot_rhs, c.d_H.p);
    }
    toc();
    if (ev) HIPCHECK(hipEventRecord(ev[1], s));
    const int l0 = c.persist_l0 >= 0 ? c.persist_l0 : P.nLevels;        // levels >= l0: one multi-level launch each way
    for (int l = 0; l < l0; l++) { cur_level = l; enqueue_factor_level(c, c.levels[l], s, tic, toc); }
    cur_level = l0;
    if (l0 < P.nLevels) { tic(K_FRONT_SMALL); launch_front_persist(c, s); toc(); }
    if (ev) HIPCHECK(hipEventRecord(ev[2], s));
    // the state update of a front's own poses rides on its back substitution (no kernel of its own); the last launch also
    // mirrors the pivot flag for the API call
    UpdArgs upd{ c.d_perm.p, gp.d_lp.p, gp.d_state.p, gp.d_dx.p, io_host ? gp.h_lp.p : nullptr, io_host ? gp.h_dx.p : nullptr, nullptr, relin ? gp.d_lp.p : nullptr };
    if (l0 < P.nLevels) {
        UpdArgs u = upd; if (l0 == 0) u.bad_out = io_host ? c.h_bad.p : nullptr;
        tic(K_BACKSOLVE);
        if (g_opt.wave_backsolve && c.p_dn_maxns <= BSW_MAX_NS)
            hipLaunchKernelGGL(k_backsolve_w, dim3(c.p_dn_n), dim3(TPB), c.p_dn_lds, s, persist_plan(c), c.d_tab.p + c.p_dn_off, c.d_pool.p, c.d_x.p, c.d_flags.p + c.flag_stride, c.d_bad.p, u);
        else
            hipLaunchKernelGGL((k_backsolve_t<true>), dim3(c.p_dn_n), dim3(TPB), c.p_dn_lds, s, persist_plan(c), c.d_tab.p + c.p_dn_off, c.d_pool.p, c.d_x.p, 0, c.d_flags.p + c.flag_stride, 1, c.d_bad.p, u);
        toc();
    }
    for (int l = l0 - 1; l >= 0; l--) {
        UpdArgs u = upd; if (l == 0) u.bad_out = io_host ? c.h_bad.p : nullptr;
        cur_level = l;
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
        const int l = i < c.k_lev.size() ? c.k_lev[i] : -1;
        if (l >= 0) {
            if (c.lev_up_ms.size() <= (size_t)l) { c.lev_up_ms.resize(l + 1, 0.0); c.lev_dn_ms.resize(l + 1, 0.0); }
            (c.k_ids[i] == K_BACKSOLVE ? c.lev_dn_ms : c.lev_up_ms)[l] += ms;
        }
    }
    c.k_ids.clear();
}

// run the numeric phase, replaying a captured hipGraph when enabled
static void run_numeric(Context &c, GraphPack &gp, bool timing, bool unary_at_lp = false, bool io_host = false, bool relin = false) {
    hipStream_t s = gp.stream;
    set_small_attr();
    rewind_epoch(c, s, 1);
    if (timing && !c.have_events) { for (auto &e : c.ev) HIPCHECK(hipEventCreate(&e)); c.have_events = true; }
    if (io_host) {
        if (g_opt.use_graph && !timing && gp.host_idx.empty()) {
            const void *key[7] = { gp.d_state.p, gp.h_state.p, gp.h_lp.p, gp.h_dx.p, c.h_bad.p, (const void *)(size_t)gp.N, (const void *)(size_t)gp.serial };
            // A graph is worth its capture, instantiation and destruction (0.3 ms together) only if the configuration comes back:
            // the first call with a new key -- every fall-back of an incremental run, every cold call -- enqueues its kernels directly.
            if (memcmp(key, c.api_key, sizeof(key)) != 0) { c.retire(c.gexec_api); memcpy(c.api_key, key, sizeof(key)); c.api_key_runs = 0; }
            if (c.api_key_runs++ == 0) { enqueue_numeric(c, gp, s, nullptr, false, false, true); return; }
            if (!c.gexec_api) {
                hipGraph_t graph = nullptr;
                HIPCHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                enqueue_numeric(c, gp, s, nullptr, false, false, true);
                HIPCHECK(hipStreamEndCapture(s, &graph));
                HIPCHECK(hipGraphInstantiate(&c.gexec_api, graph, nullptr, nullptr, 0));
                HIPCHECK(hipGraphDestroy(graph));
            }
            c.graph_stream = s;
            HIPCHECK(hipGraphLaunch(c.gexec_api, s));
        } else {
            enqueue_numeric(c, gp, s, timing ? c.ev : nullptr, false, false, true);
        }
        return;
    }
    if (g_opt.use_graph && !timing && !unary_at_lp && gp.host_idx.empty()) {   // (host-evaluated factors: staging buffers may move)
        if (!c.gexec || c.gexec_key != (const void *)gp.d_state.p || c.gexec_serial != gp.serial) {
            c.retire(c.gexec);
            hipGraph_t graph = nullptr;
            HIPCHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            enqueue_numeric(c, gp, s, nullptr, false, false, false, relin);
            HIPCHECK(hipStreamEndCapture(s, &graph));
            HIPCHECK(hipGraphInstantiate(&c.gexec, graph, nullptr, nullptr, 0));
            HIPCHECK(hipGraphDestroy(graph));
            c.gexec_key = (const void *)gp.d_state.p; c.gexec_serial = gp.serial;
        }
        c.graph_stream = s;
        HIPCHECK(hipGraphLaunch(c.gexec, s));
    } else {
        enqueue_numeric(c, gp, s, timing ? c.ev : nullptr, unary_at_lp, false, false, relin);
    }
}

static double device_chi2(GraphPack &gp) {     // chi^2 at d_state; synchronises the stream
    hipStream_t s = gp.stream;
    if (gp.F == 0) return 0;
    hipLaunchKernelGGL(k_chi2, dim3((gp.F + TPB - 1) / TPB), dim3(TPB), 0, s, gp.F, gp.d_fa.p, gp.d_fb.p, gp.d_z.p, gp.d_W.p, gp.d_state.p, gp.d_chi2f.p);
    if (gp.F > REDUCE_SPLIT) {         // (the parts live behind the F_cap per-factor terms: upload_factors)
        double *parts = gp.d_chi2f.p + gp.F_cap;
        hipLaunchKernelGGL(k_reduce_parts, dim3(REDUCE_PARTS), dim3(TPB), 0, s, gp.F, gp.d_chi2f.p, parts);
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, s, REDUCE_PARTS, parts, gp.d_scalar.p);
    } else hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, s, gp.F, gp.d_chi2f.p, gp.d_scalar.p);
    HIPCHECK(hipMemcpyAsync(gp.h_scalar.p, gp.d_scalar.p, 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    return gp.h_scalar.p[0];
}

// Options are baked into the launch tables at plan time (build_level: small / panel / big classification, tile counts,
// diagonal-block slots) AND read again when the kernels are enqueued or captured into a hipGraph.  Every change of an option
// that touches either (api_set_option bumps g_opt_epoch) therefore forces a re-plan and a re-capture on every param.
static long long g_opt_epoch = 0;
static long long launch_table_key() { return g_opt_epoch; }
// Batch-like calls, BEFORE prepare_plan: (re)derive Context::wt from the reference's own elimination order of the packed graph
// (refmodel.cpp restates it, aprilsam.c:999-1249) when the graph holds factors with an asymmetric W.  Graphs without such factors -- every
// graph the reference ships or generates -- pay one integer comparison.  Returns true when it ran the model (c.model then describes this
// batch step: the caller need not compute it again).
static void set_wt_any(Context &c, bool v) { c.wt_any = v; }
static bool orient_asymmetric(Context &c, GraphPack &gp) {
    if (gp.n_asym == 0) {
        if (c.wt_any) { c.wt.clear(); set_wt_any(c, false); c.wt_dirty = true; c.wt_serial = -1; }
        return false;
    }
    if (!gp.host_idx.empty()) fail(ERR_UNSUPPORTED, "factors with an asymmetric information matrix next to factors of foreign types: the reference's result depends on its elimination "
                                                    "order here (aprilsam.c:171), which this library only models for xyt / xytpos graphs");
    if (c.wt_serial == gp.serial && c.wt_topo == gp.topo_version && c.wt_content == gp.content_version && (int)c.wt.size() == gp.F) return false;
    const int N = gp.N, F = gp.F;
    c.model.batch(N, F, gp.h_fa.p, gp.h_fb.p);
    c.wt.assign((size_t)F, 0);
    bool any = false;
    for (int f = 0; f < F; f++) {
        const int a = gp.h_fa.p[f], b = gp.h_fb.p[f];
        if (gp.asym[f] && b >= 0 && c.model.pos[b] < c.model.pos[a]) { c.wt[f] = 1; any = true; }
    }
    set_wt_any(c, any);
    c.wt_serial = gp.serial; c.wt_topo = gp.topo_version; c.wt_content = gp.content_version; c.wt_dirty = true;
    return true;
}
// ... and AFTER it: the swap bytes of a REUSED plan follow a changed orientation (a fresh upload carries them already)
static void flush_orientation(Context &c, hipStream_t s) {
    if (!c.wt_dirty || !c.have_plan || !c.d_swap.p) return;
    fill_swap_host(c);
    HIPCHECK(hipMemcpyAsync(c.d_swap.p, c.swap_host.data(), c.swap_host.size(), hipMemcpyHostToDevice, s));
    c.wt_dirty = false;
}
// make sure plan / device buffers match the packed graph; returns true if the plan was reused
static bool prepare_plan(Context &c, GraphPack &gp, const april_graph_t *g, bool upload = true) {
    const int N = gp.N, F = gp.F;
    bool same = c.have_plan && c.patN == N && (int)c.pat.size() == 2 * F && c.plan.leaf_nodes == g_opt.leaf_nodes && c.plan_pin == g_opt.pin_last &&
                c.plan_persist == launch_table_key() && c.inc.t_first.empty();        // (a plan extended by tail fronts is only driven by inc_fast_step)
    if (same && !(c.pat_serial == gp.serial && c.pat_topo == gp.topo_version)) {      // (same pack, endpoints untouched since the last comparison: nothing to compare)
        for (int i = 0; i < F && same; i++) same = c.pat[2 * i] == gp.h_fa.p[i] && c.pat[2 * i + 1] == gp.h_fb.p[i];
    }
    c.pat_serial = gp.serial; c.pat_topo = gp.topo_version;       // (c.pat equals the packed endpoints from here on, either way)
    if (same) return true;
    c.pat.resize((size_t)2 * F);
    for (int i = 0; i < F; i++) { c.pat[2 * i] = gp.h_fa.p[i]; c.pat[2 * i + 1] = gp.h_fb.p[i]; }
    c.patN = N; c.plan_pin = g_opt.pin_last; c.plan_persist = launch_table_key();
    std::vector<double> xy((size_t)2 * N);
    for (int i = 0; i < N; i++) { xy[2 * i] = gp.h_state.p[3 * i]; xy[2 * i + 1] = gp.h_state.p[3 * i + 1]; }
    const double tb0 = now_ms();
    build_plan(c.plan, N, F, c.pat.data(), xy.data(), g_opt.leaf_nodes);
    if (g_opt.pool_guard > 0) {              // debug: a guard band behind every frontal array (upload_plan fills them, check_guard reads them)
        Plan &P = c.plan;
        const long long G = ((long long)g_opt.pool_guard + 31) & ~31ll;
        long long off = 0;
        for (int t = 0; t < P.nF; t++) { P.f_off[t] = off; off += ((((long long)P.rows(t) * P.cols(t)) + 31) & ~31ll) + G; }
        P.pool_doubles = off;
    }
    const double tb1 = now_ms();
    if (upload) upload_plan(c, gp.stream);
    if (getenv("APRILSAM_AMD_PLAN_PROFILE")) fprintf(stderr, "aprilsam_amd plan: N=%d build %.3f ms upload %.3f ms\n", N, tb1 - tb0, now_ms() - tb1);
    c.have_plan = true;
    return false;
}


// solver_shard.inc.h -- part of solver.hip.cpp (ONE translation unit: the kernels of kernels.hip.h are compiled once); included from there,
// inside namespace asam.  Contents: multi-GPU: nested-dissection subtree sharding, transports (RCCL loaded at run time, host callbacks).
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
static Registry<ShardState> g_shard;
static void drop_shard_state(const void *param) {           // (failure path; the slot lock is held)
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
    SlotLock lk(param, g);
    Context &c = ctx_for(param);
    GraphPack &gp = pack_for(g);
    pack_factors(gp, g);
    if (!gp.host_idx.empty()) return -4;
    pack_states(gp, g, false);
    c.have_plan = false;                      // the pool layout is per rank: never reuse an upload made for another layout
    orient_asymmetric(c, gp);
    prepare_plan(c, gp, g, false);
    const Plan &P = c.plan;
    { auto it = g_shard.find(param); if (it != g_shard.end()) { it->second->release(); g_shard.erase(it); } }
    g_shard.put(param, std::make_unique<ShardState>());
    auto &S = *g_shard.find(param)->second;
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
    { size_t mx = 1; for (int l = 0; l < P.nLevels; l++) mx = std::max(mx, diag_doubles(S.levels[l].n_big)); c.d_diag.need(mx); }
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
    SlotLock lk(param, nullptr);
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
    SlotLock lk(param, nullptr);
    auto it = g_shard.find(param);
    if (it == g_shard.end()) return -1;
    if (!g_rccl.load()) return -5;
    ShardState &S = *it->second;
    auto T = std::make_unique<RcclTransport>();
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    
    const ncclResult_t r = g_rccl.CommInitRank(&T->comm, S.world, id, S.rank);
    if (r != ncclSuccess) { fprintf(stderr, "aprilsam_amd: ncclCommInitRank failed: %s\n", g_rccl.GetErrorString(r)); return -6; }
    S.tr = std::move(T);
    return 0;
}
// what the attached transport is, as the communication library itself reports it: out = {kind (0 none, 1 RCCL, 2 host
// callbacks), ncclCommCount, ncclCommUserRank, ncclGetVersion code, HIP device}; path (may be null) receives the librccl
// file the symbols came from
int shard_comm_info(const april_graph_cholesky_param_t *param, long long *out, char *path, int cap) {
    SlotLock lk(param, nullptr);
    auto it = g_shard.find(param);
    if (it == g_shard.end()) return -1;
    ShardState &S = *it->second;
    out[0] = 0; out[1] = S.world; out[2] = S.rank; out[3] = 0; out[4] = physical_device(t_slot);
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
    SlotLock lk(param, nullptr);
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
    SlotLock lk(param, g);
    auto it = g_shard.find(param); auto ic = g_ctx.find(param);
    if (it == g_shard.end() || ic == g_ctx.end()) return -1;
    ShardState &S = *it->second; Context &c = *ic->second;
    if (S.world > 1 && !S.tr) return -7;
    GraphPack &gp = pack_for(g);
    const Plan &P = c.plan;
    hipStream_t s = gp.stream;
    
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
            enqueue_factor_level(c, S.levels[l], s, nop, nop0, S.d_tab.p);
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
    SlotLock lk(param, g);
    auto it = g_shard.find(param); auto ic = g_ctx.find(param);
    if (it == g_shard.end() || ic == g_ctx.end()) return -1;
    ShardState &S = *it->second; Context &c = *ic->second;
    if (S.world > 1 && !S.tr) return -7;
    GraphPack &gp = pack_for(g);
    hipStream_t s = gp.stream;
    
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
    SlotLock lk(param, g);
    auto it = g_shard.find(param); auto ic = g_ctx.find(param);
    if (it == g_shard.end() || ic == g_ctx.end()) return -1;
    GraphPack &gp = pack_for(g);
    const Plan &P = ic->second->plan; ShardState &S = *it->second;
    hipStream_t s = gp.stream;
    
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
    SlotLock lk(param, nullptr);
    auto it = g_shard.find(param);
    if (it != g_shard.end()) { it->second->release(); g_shard.erase(it); }
}


// solver_pack.inc.h -- part of solver.hip.cpp (ONE translation unit: the kernels of kernels.hip.h are compiled once); included from there,
// inside namespace asam.  Contents: the packed graph: SoA mirrors of the caller's factor and node objects in pinned host memory and in HBM, one per april_graph_t.
// ------------------------------------------------------------------------------------------------------
// packed graph (SoA, host pinned + device) — one per april_graph_t pointer
// ------------------------------------------------------------------------------------------------------
static std::atomic<long long> g_pack_serial{ 0 };
static void forget_stream(hipStream_t s);      // solver_context.inc.h: no context may record an event on a stream that is about to be destroyed
// Streams are recycled per device slot.  On this runtime hipStreamCreateWithFlags takes 8 ms (a hardware queue) and hipStreamDestroy 2 ms
// (rocprofv3 --hip-trace, round 6): a graph's first solver call paid the former, its destruction the latter -- a program that builds a
// graph per solve (tools/soak_batch.py; a front end that re-creates its graph) paid both every time.  A released pack parks its idle
// stream; the next pack of that slot takes it.
constexpr int STREAM_POOL_MAX = 4;
static std::mutex g_stream_pool_mu;
static std::vector<hipStream_t> g_stream_pool[MAX_SLOTS];
static hipStream_t take_stream(int slot) {
    {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        auto &v = g_stream_pool[slot];
        if (!v.empty()) { hipStream_t s = v.back(); v.pop_back(); return s; }
    }
    hipStream_t s = nullptr;
    HIPCHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}
static void park_stream(int slot, hipStream_t s) {
    if (hipStreamSynchronize(s) == hipSuccess) {          // (idle, and healthy: a stream that reports an error is not handed on)
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        auto &v = g_stream_pool[slot];
        if ((int)v.size() < STREAM_POOL_MAX) { v.push_back(s); return; }
    } else (void)hipGetLastError();
    (void)hipStreamDestroy(s);
}
struct GraphPack {
    int slot = 0;                      // device slot the pack's buffers and stream live on (solver.hip.cpp: SlotLock)
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
    DBuf<double> d_lp_last; bool lp_last_valid = false;        // resident loops: the linearisation point of the LAST step enqueued (d_lp itself already holds the next one, UpdArgs::lp_next)
    int F_on_device = 0;               // factors already uploaded
    int dirty_lo = 0, dirty_hi = 0;    // packed factors whose z / W changed since the last upload
    long long content_version = 0;     // bumped whenever z / W of a packed factor changed
    long long topo_version = 0;        // bumped whenever the packed endpoints (h_fa / h_fb, F) changed in any way: a pattern compared equal at (serial, topo_version) still is
    // factors whose information matrix is NOT symmetric as given (the reference's text loader fills W[1], W[2], W[5] and leaves W[3], W[6],
    // W[7] zero, examples/aprilsam_demo.c:73-75): the reference accumulates only blocks in the upper triangle of ITS elimination order with
    // W as given (aprilsam.c:171,520), so the off-diagonal block of such a factor depends on which endpoint it eliminates first
    // (solver_context.inc.h: orient_asymmetric)
    std::vector<unsigned char> asym; int n_asym = 0;
    void note_asym(int p, const double *w, bool device_factor) {
        const unsigned char a = device_factor && (w[1] != w[3] || w[2] != w[6] || w[5] != w[7]);
        if ((size_t)p >= asym.size()) asym.resize((size_t)p + 1, 0);
        n_asym += (int)a - (int)asym[p]; asym[p] = a;
    }
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
        d_fa.release(); d_fb.release(); d_z.release(); d_W.release(); d_state.release(); d_lp.release(); d_lp_last.release(); d_dx.release();
        d_chi2f.release(); d_scalar.release(); h_scalar.release(); h_hostH.release(); d_hostH.release(); d_host_idx.release(); d_upt.release();
        if (stream) { forget_stream(stream); park_stream(slot, stream); }
        stream = nullptr;
    }
};

static Registry<GraphPack> g_packs;

static int slot_of_graph(const void *g) {
    int slot = -1;
    g_packs.with(g, [&](GraphPack &gp) { slot = gp.slot; });
    return slot;
}
// the pack of a graph, on the slot of the call in progress (a pack left behind on another slot by an earlier call moves: dropped there,
// rebuilt here -- one graph is driven from one slot at a time)
static GraphPack &pack_for(const april_graph_t *g) {
    auto it = g_packs.find(g);
    if (it != g_packs.end() && it->second->slot != t_slot) {
        // The pack is released under the OLD slot's lock as well, so that no call on that slot is inside it (april_graph_chi2 or a destroy
        // found by the graph alone).  That lock is tried, not waited for without bound: this thread holds its own slot, and two threads
        // moving graphs in opposite directions must not wait for each other forever -- normally the other slot is busy with ANOTHER
        // graph's call and frees up within that call's time
        const int os = it->second->slot;
        std::unique_lock<std::mutex> old_slot(g_slot_mu[os], std::try_to_lock);
        for (const auto t0 = std::chrono::steady_clock::now(); !old_slot.owns_lock(); (void)old_slot.try_lock()) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10))
                fail(ERR_UNSUPPORTED, "graph %p was last solved on device slot %d, which other threads kept busy for 10 s: one graph is driven from one slot at a "
                     "time (the call on slot %d did nothing)", (const void *)g, os, t_slot);
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        it = g_packs.find(g);
        if (it != g_packs.end() && it->second->slot != t_slot) { it->second->release(); g_packs.erase(it); it = g_packs.end(); }
    }
    if (it == g_packs.end()) {
        auto p = std::make_unique<GraphPack>();
        p->slot = t_slot;
        p->stream = take_stream(t_slot);
        it = g_packs.emplace(g, std::move(p)).first;
    }
    return *it->second;
}
void drop_graph_pack(const april_graph_t *g) {
    SlotLock lk(nullptr, g);
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
    auto restart = [&]() { gp.topo_version++; from = 0; gp.F = 0; gp.F_on_device = 0; gp.host_idx.clear(); gp.host_evaluated = 0; gp.is_host.clear(); gp.p2g.clear(); gp.vslot.clear(); gp.g2p.assign(1, 0); gp.asym.clear(); gp.n_asym = 0; };
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
            // -1 in `b` is the internal marker of a unary entry: a graph factor with two or more nodes must name real nodes on both sides
            // (a negative second endpoint used to pass as "unary" with its node slots still naming the missing node)
            if (ents[e].a < 0 || ents[e].a >= N || ents[e].b >= N || (f->nnodes >= 2 && ents[e].b < 0))
                fail(ERR_BAD_GRAPH, "factor %d references node %d / %d of %d", i, ents[e].a, ents[e].b, N);
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
                gp.note_asym(p0, Wp, true);
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
    {   // the pinned mirrors are sized ONCE for everything this call appends (a pinned reallocation costs a quarter of a millisecond)
        size_t total = (size_t)F;
        for (int i = from; i < Fg; i++) {
            // arity is checked BEFORE anything is sized from it: a foreign factor with a garbage nnodes must end in ERR_UNSUPPORTED
            // (classify() below says so), not in a pinned allocation of k (k - 1) / 2 entries
            const int k = fs[i]->nnodes;
            total += (k >= 3 && k <= 11) ? (size_t)k * (k - 1) / 2 : 1;
        }
        gp.h_fa.need(total, true); gp.h_fb.need(total, true); gp.h_z.need(3 * total, true); gp.h_W.need(9 * total, true);
        gp.is_host.resize(total, 0); gp.p2g.resize(total); gp.vslot.resize(total);
    }
    if (from < Fg || gp.F != F) gp.topo_version++;
    for (int i = from; i < Fg; i++) {
        const april_graph_factor_t *f = fs[i];
        gp.fptr[i] = f;
        classify(f, i);
        for (int e = 0; e < ne; e++, F++) {
            gp.h_fa.p[F] = ents[e].a; gp.h_fb.p[F] = ents[e].b; gp.is_host[F] = ents[e].host; gp.p2g[F] = i;
            gp.vslot[F] = (unsigned)ents[e].slots | ((unsigned)ents[e].carry << 16);
            if (ents[e].host) {          // the device kernels see a null factor (W = 0) in its place; k_scatter_host fills its slots
                memset(gp.h_z.p + (size_t)3 * F, 0, 24); memset(gp.h_W.p + (size_t)9 * F, 0, 72);
                gp.note_asym(F, gp.h_W.p + (size_t)9 * F, false);
                gp.host_idx.push_back(F);
            } else {
                memcpy(gp.h_z.p + (size_t)3 * F, f->u.common.z, 24);
                memcpy(gp.h_W.p + (size_t)9 * F, f->u.common.W->data, 72);
                gp.note_asym(F, gp.h_W.p + (size_t)9 * F, true);
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
        gp.d_chi2f.need((size_t)gp.F_cap + REDUCE_PARTS);       // per-factor terms + the partial sums of device_chi2
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
    // (the pairs of a factor with more than two nodes share one evaluation; the guard destroys the live one on every exit, the fail() paths included)
    struct EvalGuard { april_graph_factor_eval_t *e = nullptr; ~EvalGuard() { if (e) april_graph_factor_eval_destroy(e); } } guard;
    april_graph_factor_eval_t *&e = guard.e; int e_of = -1;
    std::vector<double> JtW;
    for (int k = from; k < nh; k++) {
        const int hp = gp.host_idx[k], gi = gp.p2g[hp];
        april_graph_factor_t *f = fs[gi];
        const int x = (int)((gp.vslot[hp] >> 8) & 0xff), y = (int)(gp.vslot[hp] & 0xff), carry = (int)(gp.vslot[hp] >> 16);
        if (gi != e_of) {
            if (e) { april_graph_factor_eval_destroy(e); e = nullptr; }
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


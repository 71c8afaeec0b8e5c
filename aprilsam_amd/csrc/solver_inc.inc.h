// solver_inc.inc.h -- part of solver.hip.cpp (ONE translation unit: the kernels of kernels.hip.h are compiled once); included from there,
// inside namespace asam.  Contents: the incremental fast path: frozen base plan + tail fronts, regeneration / low-rank update of the dirty root paths (inc_prepare, inc_fast_step).
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
    I.ready = true; I.pristine = true;
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
    I.pristine = false;
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
                new_swap[f - Fold] = (unsigned char)((fb[f] >= 0 && fa[f] < fb[f] ? 1 : 0) | ((c.wt_any && f < (int)c.wt.size() && c.wt[f]) ? 2 : 0));
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
            const IncFlags nofl{ c.d_epoch.p, nullptr, nullptr, 0 };          // (one workgroup, no flags: the step counter advances all the same)
            rewind_epoch(c, s, 1);
            const int *dn = c.d_tab.p + I.tab_used; const int n_dn = solve_here ? 0 : (int)lst.size();
            const int one_nt = g_opt.inc_one_threads >= 1024 ? 1024 : g_opt.inc_one_threads >= 512 ? 512 : 256;
            if (one_nt >= 1024) hipLaunchKernelGGL(k_inc_one<1024>, dim3(1), dim3(1024), lds, s, pro, nofl, c.inl, c.dp, tstep, (const int *)nullptr, 0, dn, n_dn, c.d_pool.p, 0ll, c.d_x.p, upd1);
            else if (one_nt >= 512) hipLaunchKernelGGL(k_inc_one<512>, dim3(1), dim3(512), lds, s, pro, nofl, c.inl, c.dp, tstep, (const int *)nullptr, 0, dn, n_dn, c.d_pool.p, 0ll, c.d_x.p, upd1);
            else hipLaunchKernelGGL(k_inc_one<256>, dim3(1), dim3(256), lds, s, pro, nofl, c.inl, c.dp, tstep, (const int *)nullptr, 0, dn, n_dn, c.d_pool.p, 0ll, c.d_x.p, upd1);
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
            for (int f = Fold; f < F; f++) new_swap[f - Fold] = (unsigned char)((fb[f] >= 0 && fa[f] < fb[f] ? 1 : 0) | ((c.wt_any && f < (int)c.wt.size() && c.wt[f]) ? 2 : 0));      // (own poses: local order = id order; see add_factor below)
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
                new_swap[f - Fold] = (unsigned char)((lb >= 0 && la < lb ? 1 : 0) | ((c.wt_any && f < (int)c.wt.size() && c.wt[f]) ? 2 : 0));         // (orientation of the off-diagonal block in its contribution slot, for later re-assemblies)
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
                if (f >= Fold) new_swap[f - Fold] = (unsigned char)((lb >= 0 && la < lb ? 1 : 0) | ((c.wt_any && f < (int)c.wt.size() && c.wt[f]) ? 2 : 0));       // orientation of the off-diagonal block
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
        D.prim1 = 0; D.prim2 = 0;                      // (a regenerated front's children are re-staged: no stored children for k_assemble_big)
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
        L.all_off += sh; L.small_off += sh; L.asm_big.list_off += sh; L.asm_big.pre_off += sh;
        for (auto &x : L.syrkw) { x.list_off += sh; x.pre_off += sh; }
        for (auto &x : L.syrka) { x.list_off += sh; x.pre_off += sh; }
        for (auto &x : L.syrk1) { x.list_off += sh; x.pre_off += sh; }
        for (auto &x : L.bchain) { x.list_off += sh; x.pre_off += sh; }
        for (auto &x : L.btile) { x.list_off += sh; x.pre_off += sh; }
        L.bs_blk.list_off += sh; L.bs_blk.pre_off += sh; L.rest_off += sh;
        L.bs_gemv.list_off += sh; L.bs_gemv.pre_off += sh;
        if (l < I.nLev0) for (int t : lev_dirty[l]) {
            I.base_levels[l].solve_lds = std::max(I.base_levels[l].solve_lds, (size_t)(3 * (P.f_nsb[t] + I.cur_nub[t]) + NB + 8 + NB * (NB + 1)) * 8);
            I.base_levels[l].solve_w_lds = std::max(I.base_levels[l].solve_w_lds, backsolve_lds(3 * (P.f_nsb[t] + I.cur_nub[t]), 3 * P.f_nsb[t], true));
        }
    }
    for (int l = 0; l < nLev; l++) if (diag_doubles(dl[l].n_big) > c.d_diag.cap) return inc_fail(11);
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
        mp = mp && mp_n <= g_opt.persist_max_fronts && nFr <= c.flag_stride;
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
        if (!iu || (size_t)iu_n > c.d_upd.cap || nFr > c.flag_stride) return inc_fail(13);      // (the eligibility pass checked what iu checks: a full re-plan otherwise)
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
    if (iu && nFr > c.flag_stride) return inc_fail(13);
    rewind_epoch(c, s, 1);
    // the plan as this step's multi-level launches see it: the flags carry the step number (advanced by the step's first kernel) -- dpi: a
    // front waits for the children this step regenerates (marks, written by the prologue); dpm: a batch step on the extended plan, levels >= 1
    DevPlan dpi = c.dp; dpi.marks = c.d_marks.p;
    DevPlan dpm = c.dp; dpm.flevel = c.d_flevel.p; dpm.l0 = 1;
    int *const xfl = c.d_flags.p + c.flag_stride;
    const UpdCtx uctx = any_upd ? UpdCtx{ c.d_upd.p, c.d_wbuf.p, c.d_flags.p + (size_t)2 * c.flag_stride, gp.d_fa.p, gp.d_fb.p, gp.d_z.p, gp.d_W.p, gp.d_lp.p, gp.d_state.p } : UpdCtx{};
    if (batch) {
        PL.launch(s);
        enqueue_poison(c, s, nullptr, nFr);                // (debug option pool_poison: every front is re-factorised by a batch step)
        hipLaunchKernelGGL(k_load_states, dim3((3 * N + TPB - 1) / TPB), dim3(TPB), 0, s, 3 * N, gp.h_state.p, gp.d_state.p, gp.d_lp.p);      // l_point <- state
        hipLaunchKernelGGL((k_linearize_t<false>), dim3((F + TPB - 1) / TPB), dim3(TPB), 0, s, 0, F, (const int *)nullptr, gp.d_fa.p, gp.d_fb.p, gp.d_z.p, gp.d_W.p,
                           gp.d_lp.p, gp.d_state.p, c.d_swap.p, c.dp.slot_blk, c.dp.slot_rhs, c.d_H.p, c.d_bad.p, (const double *)nullptr, c.d_epoch.p);
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
        const IncFlags fl = (!one && iu) ? IncFlags{ c.d_epoch.p, c.d_marks.p, c.d_tab.p + iu_off, iu_n } : IncFlags{ c.d_epoch.p, nullptr, nullptr, 0 };
        if (tail_fast && !one) {                       // the refactorisation in the prologue's launch, the back substitution in launches of its own
            hipLaunchKernelGGL(k_inc_one<1024>, dim3(1), dim3(1024), tail_refactor_lds(), s, pro, fl, c.inl, c.dp, tstep, (const int *)nullptr, 0, (const int *)nullptr, 0,
                               c.d_pool.p, iu_full, c.d_x.p, UpdArgs{});
        } else if (one) {
            if (g_incprof_stamps) { c.h_kstamp.need(8 + PROF_SLOTS); memset(c.h_kstamp.p, 0, 8 * (8 + PROF_SLOTS)); pro.stamps = c.h_kstamp.p; }
            if (g_opt.inc_one_spin) {                  // completion through a word in pinned memory: the host spins instead of sleeping in hipStreamSynchronize
                if (!c.h_done.p) { c.h_done.need(16); c.h_done.p[0] = 0; }
                c.done_seq = c.done_seq >= 0x7ffffff0 ? 1 : c.done_seq + 1;
                pro.done = c.h_done.p; pro.seq = c.done_seq; c.one_wait = c.done_seq;
            }
            gp.h_out.need((size_t)3 * N);
            const UpdArgs upd1{ c.d_perm.p, gp.d_lp.p, nullptr, gp.d_dx.p, gp.h_out.p, gp.h_dx.p, c.h_bad.p };
            if (one_nt >= 1024) hipLaunchKernelGGL(k_inc_one<1024>, dim3(1), dim3(1024), one_lds, s, pro, fl, c.inl, c.dp, tstep, c.d_tab.p + iu_off, iu_n, c.d_tab.p + id_off, id_n, c.d_pool.p, iu_full, c.d_x.p, upd1, uctx);
            else if (one_nt >= 512) hipLaunchKernelGGL(k_inc_one<512>, dim3(1), dim3(512), one_lds, s, pro, fl, c.inl, c.dp, tstep, c.d_tab.p + iu_off, iu_n, c.d_tab.p + id_off, id_n, c.d_pool.p, iu_full, c.d_x.p, upd1, uctx);
            else hipLaunchKernelGGL(k_inc_one<256>, dim3(1), dim3(256), one_lds, s, pro, fl, c.inl, c.dp, tstep, c.d_tab.p + iu_off, iu_n, c.d_tab.p + id_off, id_n, c.d_pool.p, iu_full, c.d_x.p, upd1, uctx);
        } else
            hipLaunchKernelGGL(k_inc_prologue, dim3(1), dim3(1024), 0, s, pro, fl, c.inl);
    }
    if (g_opt.pool_poison > 0 && !batch && !one) {       // debug: what this step's multi-level launches hand over (the lists and descriptors arrived with the prologue's patches)
        if (iu) { enqueue_poison(c, s, c.d_tab.p + iu_off, iu_n, 1, any_upd ? c.d_upd.p : nullptr); if (any_upd) HIPCHECK(hipMemsetAsync(c.d_wbuf.p, 0xff, c.d_wbuf.cap * 8, s)); }
        if (id) enqueue_poison(c, s, c.d_tab.p + id_off, id_n, 2);
    }
    if (iu && !one) {
        const int *list = c.d_tab.p + iu_off;
        if (iu_nt >= 1024) hipLaunchKernelGGL(k_front_small<1024>, dim3(iu_n), dim3(1024), iu_lds, s, dpi, list, c.d_pool.p, c.d_H.p, c.d_bad.p, iu_full, c.d_flags.p, 1, uctx);
        else if (iu_nt >= 512) hipLaunchKernelGGL(k_front_small<512>, dim3(iu_n), dim3(512), iu_lds, s, dpi, list, c.d_pool.p, c.d_H.p, c.d_bad.p, iu_full, c.d_flags.p, 1, uctx);
        else hipLaunchKernelGGL(k_front_small<256>, dim3(iu_n), dim3(256), iu_lds, s, dpi, list, c.d_pool.p, c.d_H.p, c.d_bad.p, iu_full, c.d_flags.p, 1, uctx);
    }
    for (int l = 0; l < nLev; l++) {
        if (lev_dirty[l].empty() || iu || one) continue;
        if (mp && l >= 1) {
            if (l > 1) continue;
            const int *list = c.d_tab.p + mp_up_off;
            if (mp_nt >= 1024) hipLaunchKernelGGL(k_front_small<1024>, dim3(mp_n), dim3(1024), mp_up_lds, s, dpm, list, c.d_pool.p, c.d_H.p, c.d_bad.p, mp_full, c.d_flags.p, 1);
            else if (mp_nt >= 512) hipLaunchKernelGGL(k_front_small<512>, dim3(mp_n), dim3(512), mp_up_lds, s, dpm, list, c.d_pool.p, c.d_H.p, c.d_bad.p, mp_full, c.d_flags.p, 1);
            else hipLaunchKernelGGL(k_front_small<256>, dim3(mp_n), dim3(256), mp_up_lds, s, dpm, list, c.d_pool.p, c.d_H.p, c.d_bad.p, mp_full, c.d_flags.p, 1);
            continue;
        }
        const LevelPlan &L = dl[l];
        if (L.n_small) launch_front_small(c, L, s);
        if (L.n_big) {
            hipLaunchKernelGGL(k_assemble_big, dim3(L.asm_big.grid), dim3(TPB), L.asm_lds, s, c.dp, c.d_tab.p + L.asm_big.list_off,
                               c.d_tab.p + L.asm_big.pre_off, L.asm_big.n, c.d_pool.p, c.d_H.p);
            enqueue_big_steps(c, L, s, [](int) {}, []() {});
        }
    }
    if (mp && g_opt.wave_backsolve && mp_dn_maxns <= BSW_MAX_NS) hipLaunchKernelGGL(k_backsolve_w, dim3(mp_n), dim3(TPB), mp_dn_lds, s, c.dp, c.d_tab.p + mp_dn_off, c.d_pool.p, c.d_x.p, xfl, c.d_bad.p, UpdArgs{});
    else if (mp) hipLaunchKernelGGL((k_backsolve_t<true>), dim3(mp_n), dim3(TPB), mp_dn_lds, s, c.dp, c.d_tab.p + mp_dn_off, c.d_pool.p, c.d_x.p, 0, xfl, 1, c.d_bad.p, UpdArgs{});
    // incremental steps: the state update (state = l_point + dx, pinned mirrors of state / dx / failure record) rides on the
    // back substitution of the front that owns the pose -- every visited pose lives in a front of this sweep -- instead of
    // a launch of its own over all poses
    // (the device states are NOT touched: d_state / d_lp keep mirroring the host objects, the new states go to a pinned buffer
    // of their own -- see pack_states_diff)
    gp.h_out.need((size_t)3 * N);
    const UpdArgs upd = batch ? UpdArgs{} : UpdArgs{ c.d_perm.p, gp.d_lp.p, nullptr, gp.d_dx.p, gp.h_out.p, gp.h_dx.p, c.h_bad.p };
    bool rode = one;
    if (id && !one) {
        if (g_opt.wave_backsolve && id_maxns <= BSW_MAX_NS) hipLaunchKernelGGL(k_backsolve_w, dim3(id_n), dim3(TPB), id_lds, s, c.dp, c.d_tab.p + id_off, c.d_pool.p, c.d_x.p, xfl, c.d_bad.p, upd);
        else hipLaunchKernelGGL((k_backsolve_t<true>), dim3(id_n), dim3(TPB), id_lds, s, c.dp, c.d_tab.p + id_off, c.d_pool.p, c.d_x.p, 0, xfl, 1, c.d_bad.p, upd);
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
// debug option pool_guard, after a synchronised step: did any kernel write into a guard band?
static void check_guard(Context &c, hipStream_t s) {
    if (c.n_guard <= 0 || !c.d_pool.p) return;
    int cnt[2] = { 0, 0 };
    hipLaunchKernelGGL(k_guard, dim3(c.n_guard), dim3(TPB), 0, s, 1, c.d_guard.p, c.n_guard, c.guard_len, c.d_pool.p, c.d_guard_cnt.p);
    HIPCHECK(hipMemcpyAsync(cnt, c.d_guard_cnt.p, 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    if (cnt[0]) fail(ERR_GUARD, "pool_guard: %d words of the guard bands were overwritten (one of them: the band behind front %d of %d)", cnt[0], cnt[1], c.n_guard);
}
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


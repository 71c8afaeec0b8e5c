// solver_calls.inc.h -- part of solver.hip.cpp (ONE translation unit: the kernels of kernels.hip.h are compiled once); included from there,
// inside namespace asam.  Contents: the reference entry points: april_graph_cholesky (batch_impl), april_graph_cholesky_inc (inc_impl), april_graph_cholesky_inc_solver, april_graph_chi2.
// ------------------------------------------------------------------------------------------------------
// one batch Gauss-Newton step through the reference API (aprilsam.c:87-375)
// ------------------------------------------------------------------------------------------------------
// APRILSAM_AMD_INC_PROFILE=1: host wall-clock split of the batch calls that made a new plan (the fall-backs of an incremental run), printed at exit
static double g_fbprof[6] = { 0 }; static long long g_fbprof_n = 0;
static const bool g_fbprof_on = [] { const char *e = getenv("APRILSAM_AMD_INC_PROFILE"); return e && (*e == '1' || *e == '2'); }();
struct FbProfPrint { ~FbProfPrint() { if (g_fbprof_on && g_fbprof_n) fprintf(stderr, "aprilsam_amd inc profile, batch calls with a new plan: %lld, ms per call: enqueue %.3f model %.3f l_point walk %.3f retired graphs %.3f wait for the GPU %.3f\n",
    g_fbprof_n, g_fbprof[0] / g_fbprof_n, g_fbprof[1] / g_fbprof_n, g_fbprof[2] / g_fbprof_n, g_fbprof[3] / g_fbprof_n, g_fbprof[4] / g_fbprof_n); } };
static FbProfPrint g_fbprof_print;
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
                     c.plan.leaf_nodes == g_opt.leaf_nodes && c.plan_pin == g_opt.pin_last && g_opt.use_graph && !param->show_timing && !c.no_speculation &&
                     gp.n_asym == 0 && !c.wt_any;       // (asymmetric W: the factors are read first -- their orientation bits depend on what they hold)
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
    // factors with an information matrix that is not symmetric as given: which off-diagonal block the reference accumulates depends on its own
    // elimination order (aprilsam.c:171); no-op for every other graph
    const bool model_by_orientation = orient_asymmetric(c, gp);
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
        bool ext = g_opt.batch_extend && !timing && c.have_plan && c.inc.ready && gp.host_idx.empty() && gp.n_asym == 0 && !c.wt_any && N >= c.patN && F >= patF &&
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
        flush_orientation(c, gp.stream);
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
    const bool fbp = g_fbprof_on && !hybrid && !speculate && !reused;
    const double f0 = fbp ? now_ms() : 0;
    // while the GPU works: a param that is used incrementally needs the reference's elimination tree of THIS batch step for
    // its next april_graph_cholesky_inc (refmodel.cpp: the reference's own min-degree order, 0.3-0.5 ms of integer work at M3500's sizes).
    // (Round 6 ran it on a thread of its own beside the planner / the extension of the plan: the calls that make a new plan gained 0.25 ms
    // each, the others lost as much to the thread's start and its cold caches -- the demo's 50 fall-backs 26.9 against 26.6 ms: not kept.)
    bool model_ready = model_by_orientation;
    if (c.used_inc && gp.host_idx.empty() && !model_ready) { c.model.batch(N, F, gp.h_fa.p, gp.h_fb.p); model_ready = true; }      // (not for params that only ever see batch calls)
    const double f1 = fbp ? now_ms() : 0;
    // ... and the part of the write-back that does not wait for the result: every node is re-linearised at the state it came
    // in with before anything is solved (aprilsam.c:131-135: l_point = state, whatever the factorisation says later), UID = index
    // (aprilsam.c:628).  The walk also pulls the node objects into the cache for the second half below.
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    for (int i = N - 1; i >= 0; i--) { april_graph_node_t *n = ns[i]; n->UID = i; memcpy(n->l_point, gp.h_state.p + (size_t)3 * i, 24); }
    const double f2 = fbp ? now_ms() : 0;
    c.reap_retired();                                  // (graphs of earlier plans: destroyed under the GPU's work)
    const double f3 = fbp ? now_ms() : 0;
    HIPCHECK(hipStreamSynchronize(gp.stream));
    const double t4 = now_ms();
    if (fbp) { g_fbprof[0] += f0 - t3; g_fbprof[1] += f1 - f0; g_fbprof[2] += f2 - f1; g_fbprof[3] += f3 - f2; g_fbprof[4] += t4 - f3; g_fbprof_n++; }
    check_bad(c);
    check_guard(c, gp.stream);
    c.st.error_code = 0;
    if (c.st.not_spd) {
        c.model.valid = false;
        static bool warned = false;
        if (!warned) { fprintf(stderr, "aprilsam_amd: information matrix not positive definite; node states left untouched\n"); warned = true; }
    } else {
        // write back: state / delta_X where not NaN-skipped (l_point and UID went in above).  Three dependent loads per node (pointer array ->
        // node object -> its state / delta_X arrays, each a heap block of its own): the objects a few nodes ahead are requested early
        for (int i = N - 1; i >= 0; i--) {                                   // aprilsam.c:311-315 order
            if (i >= 8) { const april_graph_node_t *p = ns[i - 8]; __builtin_prefetch(p->state, 1); __builtin_prefetch(p->delta_X, 1); }
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
        if (!hybrid && !(reused && c.inc.ready && c.inc.pristine)) inc_prepare(c);      // (an extended plan keeps its base + tail bookkeeping; a warm call on an untouched plan its tables: rebuilding them was 40 % of the call's time behind the stream sync)
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
        SlotLock lk(param, g);
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
    std::vector<float> regen;                         // ... and the fronts it regenerated (-1: re-planned)
    std::vector<std::array<float, 4>> kst;            // =2: phases of k_inc_one in us (patches, linearise, fronts, back substitution)
    std::vector<std::array<float, 10>> fst;           //     ... and of its last front
    IncProf() { const char *e = getenv("APRILSAM_AMD_INC_PROFILE"); on = e && (*e == '1' || *e == '2'); }
    ~IncProf() {
        if (!on || !n) return;
        if (const char *path = getenv("APRILSAM_AMD_INC_PROFILE_DUMP")) {       // per step: 7 floats (the six phases + total) and the fronts it regenerated, for tools/inc_phases.py
            if (FILE *fp = fopen(path, "wb")) { for (size_t i = 0; i < steps.size(); i++) { fwrite(steps[i].data(), 4, 7, fp); const float r = i < regen.size() ? regen[i] : -1.f; fwrite(&r, 4, 1, fp); } fclose(fp); }
        }
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
    SlotLock lk(param, g);
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
    if (gp.n_asym > 0 || c.wt_any) {          // new factors with an asymmetric W: oriented by the positions they enter at (aprilsam.c:393-396,520)
        if (!gp.host_idx.empty()) fail(ERR_UNSUPPORTED, "factors with an asymmetric information matrix next to factors of foreign types (see orient_asymmetric)");
        c.wt.resize((size_t)F, 0);
        for (int f = c.inc_F; f < F; f++) {
            const int a = gp.h_fa.p[f], b = gp.h_fb.p[f];
            c.wt[f] = (f < (int)gp.asym.size() && gp.asym[f] && b >= 0 && c.model.pos[b] < c.model.pos[a]) ? 1 : 0;
            if (c.wt[f]) set_wt_any(c, true);
        }
    }
    std::vector<RefModel::Visit> &visits = c.visits;
    const bool partial = c.model.naffected <= 5;     // aprilsam.c:755: otherwise the whole tree is walked
    // structural: which poses the reference's solve_node touches.  A partial walk's list decides what the GPU back-substitutes: now.  A
    // full walk's list (every pose, in the reference's tree order: 14 us on M3500) is only needed when the numbers are back: it is
    // made while the GPU works, below
    if (partial) c.model.plan_visit(visits); else visits.clear();
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
    if (!partial) c.model.plan_visit(visits);        // (the full walk's list, under the GPU's work)
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
    check_guard(c, gp.stream);
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
        g_incprof.regen.push_back(reused ? (float)c.st.reserved0 : -1.f);
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
        SlotLock lk(param, g);
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
    SlotLock lk(nullptr, g);
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
            april_graph_factor_eval_t *e = fs[gi]->eval ? fs[gi]->eval(fs[gi], g, nullptr) : nullptr;
            if (!e) fail(ERR_BAD_GRAPH, "factor %d: eval() returned no evaluation (aprilsam.h:75-89)", gi);      // (never takes the process down: errors.h)
            chi2 += e->chi2;
            april_graph_factor_eval_destroy(e);
        }
    }
    return chi2;
}


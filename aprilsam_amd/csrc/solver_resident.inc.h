// solver_resident.inc.h -- part of solver.hip.cpp (ONE translation unit: the kernels of kernels.hip.h are compiled once); included from there,
// inside namespace asam.  Contents: the device-resident driver API, kernel / level profiles, stage-level parity exports.
// ------------------------------------------------------------------------------------------------------
// device-resident driver API: states never leave HBM between Gauss-Newton steps
// ------------------------------------------------------------------------------------------------------
static int resident_begin_impl(april_graph_t *g, april_graph_cholesky_param_t *param);
int resident_begin(april_graph_t *g, april_graph_cholesky_param_t *param) { return guarded_rc(param, g, [&] { return resident_begin_impl(g, param); }); }
static int resident_begin_impl(april_graph_t *g, april_graph_cholesky_param_t *param) {
    if (zsize(g->nodes) == 0 || zsize(g->factors) == 0) return -1;
    ensure_device();
    SlotLock lk(param, g);
    Context &c = ctx_for(param);
    GraphPack &gp = pack_for(g);
    pack_factors(gp, g);
    if (!gp.host_idx.empty()) return -4;          // host-evaluated factors need the host in the loop: use april_graph_cholesky
    pack_states(gp, g, false);
    gp.lp_last_valid = false;
    orient_asymmetric(c, gp);
    const bool reused = prepare_plan(c, gp, g);
    flush_orientation(c, gp.stream);
    upload_factors(gp);
    set_lambda(c, gp, param->tikhanov);
    if (!c.have_events) { for (auto &e : c.ev) HIPCHECK(hipEventCreate(&e)); c.have_events = true; }
    for (int k = 0; k < NKERN; k++) { c.k_ms[k] = 0; c.k_calls[k] = 0; }
    c.lev_up_ms.clear(); c.lev_dn_ms.clear();
    c.st.n_nodes = gp.N; c.st.n_factors = gp.F; c.st.symbolic_reused = reused; c.st.not_spd = 0;
    HIPCHECK(hipStreamSynchronize(gp.stream));
    return 0;
}
// enqueue n iterations.  mode 0: asynchronous (hipGraph replay when enabled), returns at once;
// mode 1: every kernel bracketed by HIP events on the solver stream, synchronises after each iteration.
static int resident_steps_impl(april_graph_t *g, april_graph_cholesky_param_t *param, int n, int mode);
int resident_steps(april_graph_t *g, april_graph_cholesky_param_t *param, int n, int mode) { return guarded_rc(param, g, [&] { return resident_steps_impl(g, param, n, mode); }); }
static int resident_steps_impl(april_graph_t *g, april_graph_cholesky_param_t *param, int n, int mode) {
    SlotLock lk(param, g);
    auto it = g_ctx.find(param);
    if (it == g_ctx.end() || !it->second->have_plan) return -1;
    Context &c = *it->second;
    GraphPack &gp = pack_for(g);
    hipStream_t s = gp.stream;
    const int N = gp.N;
    set_small_attr();
    gp.mirror_sync = false;                           // (states move on the device only)
    // every node is re-linearised at the state it has when a batch step begins (aprilsam.c:131-135): once here, and from then on the state
    // update of step i leaves the new states in the l_points as well (UpdArgs::lp_next) -- no 84 KB copy between two steps of the loop
    // (round 5: 0.232 -> 0.225 ms per M3500 iteration)
    // What the node objects must hold when the loop ends is the linearisation point of the LAST step -- the reference's l_point after K
    // calls of april_graph_cholesky, the point the factorisation kept for april_graph_cholesky_inc was made at (aprilsam.c:131-135, 508-542) --
    // and d_lp has moved on by then: one copy per CALL (not per step), before its last step.
    HIPCHECK(hipMemcpyAsync(gp.d_lp.p, gp.d_state.p, (size_t)24 * N, hipMemcpyDeviceToDevice, s));
    for (int i = 0; i < n; i++) {
        if (i == n - 1) {
            gp.d_lp_last.need((size_t)3 * N);
            HIPCHECK(hipMemcpyAsync(gp.d_lp_last.p, gp.d_lp.p, (size_t)24 * N, hipMemcpyDeviceToDevice, s));
            gp.lp_last_valid = true;
        }
        if (mode == 1) {
            enqueue_numeric(c, gp, s, nullptr, false, true, false, true);
            HIPCHECK(hipStreamSynchronize(s));
            collect_kernel_times(c);
        } else {
            run_numeric(c, gp, false, false, false, true);
        }
    }
    return 0;
}
static int resident_sync_impl(april_graph_t *g, april_graph_cholesky_param_t *param);
int resident_sync(april_graph_t *g, april_graph_cholesky_param_t *param) { return guarded_rc(param, g, [&] { return resident_sync_impl(g, param); }); }
static int resident_sync_impl(april_graph_t *g, april_graph_cholesky_param_t *param) {
    SlotLock lk(param, g);
    auto it = g_ctx.find(param);
    if (it == g_ctx.end()) return -1;
    Context &c = *it->second;
    GraphPack &gp = pack_for(g);
    HIPCHECK(hipMemcpyAsync(c.h_bad.p, c.d_bad.p, 16, hipMemcpyDeviceToHost, gp.stream));
    HIPCHECK(hipStreamSynchronize(gp.stream));
    check_bad(c);
    check_guard(c, gp.stream);
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
    SlotLock lk(nullptr, g);
    return device_chi2(pack_for(g));
}
static int resident_end_impl(april_graph_t *g, april_graph_cholesky_param_t *param);
int resident_end(april_graph_t *g, april_graph_cholesky_param_t *param) { return guarded_rc(param, g, [&] { return resident_end_impl(g, param); }); }
static int resident_end_impl(april_graph_t *g, april_graph_cholesky_param_t *param) {
    SlotLock lk(param, g);
    auto it = g_ctx.find(param);
    if (it == g_ctx.end()) return -1;
    Context &c = *it->second;
    GraphPack &gp = pack_for(g);
    hipStream_t s = gp.stream;
    const int N = gp.N, F = gp.F;
    HIPCHECK(hipMemcpyAsync(gp.h_state.p, gp.d_state.p, (size_t)24 * N, hipMemcpyDeviceToHost, s));
    // l_point: where the last step was linearised (resident_steps parked it); d_lp follows, so that mirror and device agree again
    if (gp.lp_last_valid && gp.d_lp_last.cap >= (size_t)3 * N) HIPCHECK(hipMemcpyAsync(gp.d_lp.p, gp.d_lp_last.p, (size_t)24 * N, hipMemcpyDeviceToDevice, s));
    gp.lp_last_valid = false;
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
    record_unary_points(gp, 0, F, gp.h_lp.p);                // (unary factors were last linearised at the last step's l_points)
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
// per level of the assembly tree: HIP-event time of the instrumented passes since resident_begin (factorisation incl. assembly /
// back substitution), fronts, fronts on the multi-workgroup path, widest own part (scalar columns), sum c_j^2 flops.  A
// multi-level launch is booked on its first level.  out: 6 doubles per level; returns the number of levels.
int level_profile(const april_graph_cholesky_param_t *param, double *out, int cap_levels) {
    SlotLock lk(param, nullptr);
    auto it = g_ctx.find(param);
    if (it == g_ctx.end() || !it->second->have_plan) return -1;
    Context &c = *it->second;
    const Plan &P = c.plan;
    for (int l = 0; l < P.nLevels && l < cap_levels; l++) {
        double *o = out + 6 * l;
        o[0] = l < (int)c.lev_up_ms.size() ? c.lev_up_ms[l] : 0.0; o[1] = l < (int)c.lev_dn_ms.size() ? c.lev_dn_ms[l] : 0.0;
        o[2] = P.lev_ptr[l + 1] - P.lev_ptr[l]; o[3] = l < (int)c.levels.size() ? c.levels[l].n_big : 0; o[4] = 0; o[5] = 0;
        for (int k = P.lev_ptr[l]; k < P.lev_ptr[l + 1]; k++) {
            const int t = P.lev_fronts[k];
            const double ns = 3.0 * P.f_nsb[t], nu = 3.0 * P.f_nub[t];
            o[4] = std::max(o[4], ns);
            o[5] += ns * (nu + 1) * (nu + 1) + ns * ns * (nu + 1) + ns * ns * ns / 3.0;       // ~ sum_j c_j^2
        }
    }
    return P.nLevels;
}
// debug option pool_guard checking itself: declares ONE extra band that lies inside the first frontal array (which the last step
// wrote) and runs the check, which must then report ERR_GUARD; the param's context is dropped by the failure path as for any error
int debug_guard_selftest(const april_graph_cholesky_param_t *param) {
    return guarded_rc(param, nullptr, [&]() -> int {
        SlotLock lk(param, nullptr);
        auto it = g_ctx.find(param);
        if (it == g_ctx.end() || !it->second->have_plan || it->second->n_guard <= 0) return -1;
        Context &c = *it->second;
        const long long inside = c.plan.f_off[0];
        HIPCHECK(hipMemcpy(c.d_guard.p, &inside, 8, hipMemcpyHostToDevice));
        check_guard(c, nullptr);
        return 0;
    });
}
int kernel_profile(const april_graph_cholesky_param_t *param, double *ms, long long *calls, double *flops, double *bytes, const char **names) {
    SlotLock lk(param, nullptr);
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
            // multiplies the upper halves of its diagonal tiles: executed, not algorithmic, not counted).  The panel kernels
            // (k_block_chain + k_block_solve) get the rest of the front's sum c_j^2: diagonal blocks, row solves, in-block products.
            const double Rv = R - 2;
            double fsy = 0;
            auto trapezoid = [&](double c_lo, double c_hi) { const double n = c_hi - c_lo; return n <= 0 ? 0.0 : n * Rv - (c_lo + c_hi - 1) * n / 2; };   // elements (i >= j) of columns [c_lo, c_hi), rows < Rv
            const int steps = ((int)ns + NB - 1) / NB;
            for (int o = 0; o * OBP < steps; o++) {
                const double k_lo = (double)o * OBP * NB, k_hi = std::min<double>(ns, (double)(o + 1) * OBP * NB);
                fsy += 2.0 * (k_hi - k_lo) * trapezoid(k_hi, C);
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
    SlotLock lk(param, g);
    Context &c = ctx_for(param);
    GraphPack &gp = pack_for(g);
    pack_factors(gp, g);
    if (!gp.host_idx.empty()) return -4;
    pack_states(gp, g, false);
    orient_asymmetric(c, gp);
    prepare_plan(c, gp, g);
    flush_orientation(c, gp.stream);
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
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o[9 + i * 3 + j] = P.fac_swap[f] ? b1[j * 3 + i] : b1[i * 3 + j];      // (rows a, columns b: for a factor with an asymmetric W that the reference enters as (b, a), the transpose of its J_b^T W J_a)
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
        SlotLock lk(param, nullptr);
        auto it = g_ctx.find(param);
        if (it == g_ctx.end() || !it->second->d_prof.p) return -1;
        HIPCHECK(hipDeviceSynchronize());
        int n = std::min(n_fronts, it->second->plan.nF);
        HIPCHECK(hipMemcpy(out, it->second->d_prof.p, (size_t)8 * PROF_SLOTS * n, hipMemcpyDeviceToHost));
        return n;
    });
}

// capi.cpp — the C-ABI entry points of libaprilsam_amd.so (include/aprilsam_amd.h PART 2 and 4).
// Same names, argument meaning and error behaviour as the reference functions they replace; each one
// forwards to the HIP runtime in solver.hip.cpp.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/aprilsam_amd.h"
#include "plan.h"
#include "errors.h"
#include "refmodel.h"
#include "solver.h"

extern "C" {

// replaces aprilsam.c:33-39 (the banner the reference's example programs print first, aprilsam.h:44)
void APRILSAM_VERSION(void) {
    printf("================\n");
    printf("| APRILSAM 0.0 |   solver: aprilsam_amd (MI355X / gfx950)\n");
    printf("================\n");
    printf("\n");
}

// replaces aprilsam.c:45-64
void april_graph_cholesky_param_init(april_graph_cholesky_param_t *param) {
    asam::drop_context(param);                 // a re-initialised param starts with no solver state ...
    asam::unbind_param(param);                 // ... on the default device slot
    memset(param, 0, sizeof(*param));
    param->tikhanov = 0.0001;
    param->nreordering = 1;
    asam::warm_up();                           // once per process: the HIP runtime's lazy initialisations happen here, not inside the first solver calls
}

// replaces aprilsam.c:66-85 — frees the owned arrays and param itself (the caller heap-allocates it)
void april_graph_cholesky_param_destory(april_graph_cholesky_param_t *param) {
    if (!param) return;
    asam::drop_context(param);
    asam::unbind_param(param);
    free(param->delta_x); free(param->B); free(param->y); free(param->ordering);
    // chol / A / tr are never set by this library (reference-owned CPU state)
    free(param);
}

// replaces aprilsam.c:87-375
void april_graph_cholesky(april_graph_t *graph, april_graph_cholesky_param_t *param) { asam::batch_step(graph, param); }

// replaces aprilsam.c:377-576
void april_graph_cholesky_inc(april_graph_t *graph, april_graph_cholesky_param_t *param) { asam::inc_step(graph, param); }

// replaces aprilsam.c:578-597
void april_graph_cholesky_inc_solver(april_graph_t *graph, april_graph_cholesky_param_t *param, int *idxs) {
    (void)idxs;                                // the unknown numbering lives in the side context
    asam::inc_solve_only(graph, param);
}

// replaces april_graph.c:79-98
double april_graph_chi2(april_graph_t *graph) { return asam::graph_chi2(graph); }

int aprilsam_amd_device_count(void) { return asam::api_device_count(); }
int aprilsam_amd_set_device(int device) { return asam::api_set_device(device); }
int aprilsam_amd_param_set_device(const april_graph_cholesky_param_t *param, int slot) { return asam::api_param_set_device(param, slot); }
int aprilsam_amd_param_get_device(const april_graph_cholesky_param_t *param) { return asam::api_param_get_device(param); }
int aprilsam_amd_set_option(const char *name, double value) { return asam::api_set_option(name, value); }
int aprilsam_amd_debug_guard_selftest(const april_graph_cholesky_param_t *param) { return asam::debug_guard_selftest(param); }
int aprilsam_amd_get_option(const char *name, double *value) { return asam::api_get_option(name, value); }
int aprilsam_amd_get_stats(const april_graph_cholesky_param_t *param, aprilsam_amd_stats_t *out) {
    return asam::get_stats(param, out) ? 0 : -1;
}
int aprilsam_amd_batch_resident(april_graph_t *graph, april_graph_cholesky_param_t *param, int iters, double *chi2_out, double *ms_out) {
    return asam::batch_resident(graph, param, iters, chi2_out, ms_out);
}
int aprilsam_amd_resident_begin(april_graph_t *graph, april_graph_cholesky_param_t *param) { return asam::resident_begin(graph, param); }
int aprilsam_amd_resident_steps(april_graph_t *graph, april_graph_cholesky_param_t *param, int n, int mode) { return asam::resident_steps(graph, param, n, mode); }
int aprilsam_amd_resident_sync(april_graph_t *graph, april_graph_cholesky_param_t *param) { return asam::resident_sync(graph, param); }
double aprilsam_amd_resident_chi2(april_graph_t *graph) { return asam::resident_chi2(graph); }
int aprilsam_amd_resident_end(april_graph_t *graph, april_graph_cholesky_param_t *param) { return asam::resident_end(graph, param); }
int aprilsam_amd_level_profile(const april_graph_cholesky_param_t *param, double *out6, int cap_levels) { return asam::level_profile(param, out6, cap_levels); }
int aprilsam_amd_kernel_profile(const april_graph_cholesky_param_t *param, double *ms, long long *calls, double *flops, double *bytes, const char **names) {
    return asam::kernel_profile(param, ms, calls, flops, bytes, names);
}
int aprilsam_amd_debug_stage(april_graph_t *graph, april_graph_cholesky_param_t *param, int what, double *out) { return asam::debug_stage(graph, param, what, out); }
int aprilsam_amd_debug_front_times(const april_graph_cholesky_param_t *param, long long *out, int n_fronts) { return asam::debug_front_times(param, out, n_fronts); }
// host logic, no GPU: the reference's elimination order (aprilsam.c:999-1249 restated) and block elimination
// tree for a graph given as factor endpoint arrays; out_order / out_parent: n_nodes ints each
int aprilsam_amd_reference_order(int n_nodes, int n_factors, const int *fa, const int *fb, int *out_order, int *out_parent) {
    asam::RefModel m;
    m.batch(n_nodes, n_factors, fa, fb);
    memcpy(out_order, m.ord.data(), sizeof(int) * (size_t)n_nodes);
    if (out_parent) memcpy(out_parent, m.parent.data(), sizeof(int) * (size_t)n_nodes);
    return 0;
}
int aprilsam_amd_shard_begin(april_graph_t *graph, april_graph_cholesky_param_t *param, int rank, int world) { return asam::shard_begin(graph, param, rank, world); }
long long aprilsam_amd_shard_info(const april_graph_cholesky_param_t *param, int what, long long *out, long long cap) { return asam::shard_info(param, what, out, cap); }
int aprilsam_amd_shard_comm_unique_id(char *out128) { return asam::shard_comm_unique_id(out128); }
int aprilsam_amd_shard_comm_init_rccl(april_graph_cholesky_param_t *param, const char *id128) { return asam::shard_comm_init_rccl(param, id128); }
int aprilsam_amd_shard_comm_init_host(april_graph_cholesky_param_t *param, const aprilsam_amd_host_comm_t *cb) { return asam::shard_comm_init_host(param, cb); }
int aprilsam_amd_shard_comm_info(const april_graph_cholesky_param_t *param, long long *out5, char *rccl_path, int cap) { return asam::shard_comm_info(param, out5, rccl_path, cap); }
int aprilsam_amd_shard_iterate(april_graph_t *graph, april_graph_cholesky_param_t *param, int n) { return asam::shard_iterate(graph, param, n); }
int aprilsam_amd_shard_gather_states(april_graph_t *graph, april_graph_cholesky_param_t *param) { return asam::shard_gather_states(graph, param); }
double aprilsam_amd_shard_chi2(april_graph_t *graph, april_graph_cholesky_param_t *param) { return asam::shard_chi2(graph, param); }
void aprilsam_amd_shard_end(april_graph_cholesky_param_t *param) { asam::shard_end(param); }

// test handle on the bookkeeping model (host logic only)
void *aprilsam_amd_refmodel_create(void) { return new asam::RefModel(); }
void aprilsam_amd_refmodel_destroy(void *m) { delete (asam::RefModel *)m; }
void aprilsam_amd_refmodel_batch(void *m, int n_nodes, int n_factors, const int *fa, const int *fb) { ((asam::RefModel *)m)->batch(n_nodes, n_factors, fa, fb); }
int aprilsam_amd_refmodel_inc_begin(void *m, int n_nodes, int n_factors, const int *fa, const int *fb) {
    asam::RefModel *M = (asam::RefModel *)m;
    M->inc_begin(n_nodes, n_factors, fa, fb);
    return M->naffected;
}
// visited[i]: 0 = untouched, 1 = delta_X only, 2 = updated.  Returns start_over after the traversal.
int aprilsam_amd_refmodel_solve_visit(void *m, const double *x, double dxy, double dth, int *visited) {
    asam::RefModel *M = (asam::RefModel *)m;
    for (int i = 0; i < M->N; i++) visited[i] = 0;
    M->solve_visit(x, dxy, dth, [&](int n, bool upd) { visited[n] = upd ? 2 : 1; });
    return M->start_over;
}
void aprilsam_amd_refmodel_get(void *m, int *parent, int *changed, int *relin) {
    asam::RefModel *M = (asam::RefModel *)m;
    for (int i = 0; i < M->N; i++) { if (parent) parent[i] = M->parent[i]; if (changed) changed[i] = M->changed[i]; if (relin) relin[i] = M->relin[i]; }
}
int aprilsam_amd_refmodel_check(void *m) { return ((asam::RefModel *)m)->check_tree(); }
int aprilsam_amd_last_error(char *msg, int cap) { return asam::get_last_error(msg, cap); }
void aprilsam_amd_clear_error(void) { asam::clear_last_error(); }
int aprilsam_amd_selftest(void) { return asam::selftest(); }
const char *aprilsam_amd_version(void) { return "aprilsam_amd 0.1 (gfx950, multifrontal FP64)"; }
void aprilsam_amd_free(void *p) { free(p); }

// ---- host-logic introspection: ordering + symbolic plan, no GPU involved -------------------------------
struct aprilsam_amd_plan { asam::Plan P; };

aprilsam_amd_plan_t *aprilsam_amd_plan_create(int n_nodes, int n_factors, const int *factor_nodes, const double *xy, int leaf_nodes) {
    aprilsam_amd_plan *pl = new aprilsam_amd_plan();
    try {
        asam::build_plan(pl->P, n_nodes, n_factors, factor_nodes, xy, leaf_nodes > 0 ? leaf_nodes : asam::g_opt.leaf_nodes);
    } catch (const asam::SolverError &e) {       // errors.h: NULL + aprilsam_amd_last_error
        asam::set_last_error(e.code, e.msg);
        fprintf(stderr, "aprilsam_amd: ERROR %d: %s\n", e.code, e.msg.c_str());
        delete pl;
        return nullptr;
    } catch (const std::exception &e) {           // std::bad_alloc from the planner or its pool tasks must not cross the C ABI either
        asam::set_last_error(asam::ERR_INTERNAL, e.what());
        fprintf(stderr, "aprilsam_amd: ERROR %d: %s\n", (int)asam::ERR_INTERNAL, e.what());
        delete pl;
        return nullptr;
    }
    return pl;
}
void aprilsam_amd_plan_destroy(aprilsam_amd_plan_t *plan) { delete plan; }

// ownership / exchange lists of a `world`-rank sharded run of this plan (host logic only; same code shard_begin uses)
long long aprilsam_amd_shard_plan(const aprilsam_amd_plan_t *plan, int world, int what, long long *out, long long cap) {
    if (!plan || world < 1) return -1;
    std::vector<int> owner; std::vector<char> top; std::vector<long long> xfer, bcast, v;
    asam::shard_map(plan->P, world, owner, top, xfer, bcast);
    if (what == 1) v = xfer; else if (what == 2) v = bcast; else if (what == 3) v.assign(owner.begin(), owner.end());
    else if (what == 4) v = asam::shard_critical_path(plan->P, world, owner, top); else return -1;
    if (out) for (long long i = 0; i < (long long)v.size() && i < cap; i++) out[i] = v[i];
    return (long long)v.size();
}

long long aprilsam_amd_plan_query(const aprilsam_amd_plan_t *plan, const char *what, long long **out) {
    std::string k(what);
    if (k.compare(0, 3, "bd_") == 0 || k.compare(0, 3, "rd_") == 0) asam::build_gather_lists(const_cast<asam::Plan &>(plan->P));
    const asam::Plan &P = plan->P;
    std::vector<long long> v;
    auto from = [&](const auto &a) { v.assign(a.begin(), a.end()); };
    if (k == "perm") from(P.perm);
    else if (k == "pos") from(P.pos);
    else if (k == "front_first") from(P.f_first);
    else if (k == "front_nsb") from(P.f_nsb);
    else if (k == "front_nub") from(P.f_nub);
    else if (k == "front_parent") from(P.f_parent);
    else if (k == "front_level") from(P.f_level);
    else if (k == "front_rows_ptr") from(P.f_rows_ptr);
    else if (k == "front_rows") from(P.f_rows);
    else if (k == "front_rel") from(P.f_rel);
    else if (k == "front_off") from(P.f_off);
    else if (k == "ch_ptr") from(P.ch_ptr);
    else if (k == "ch_idx") from(P.ch_idx);
    else if (k == "factor_front") from(P.fac_front);
    else if (k == "factor_la") from(P.fac_la);
    else if (k == "factor_lb") from(P.fac_lb);
    else if (k == "factor_swap") from(P.fac_swap);
    else if (k == "bd_front_ptr") from(P.bd_front_ptr);
    else if (k == "bd_row") from(P.bd_row);
    else if (k == "bd_col") from(P.bd_col);
    else if (k == "bd_src_ptr") from(P.bd_src_ptr);
    else if (k == "bd_src") from(P.bd_src);
    else if (k == "rd_front_ptr") from(P.rd_front_ptr);
    else if (k == "rd_col") from(P.rd_col);
    else if (k == "rd_src_ptr") from(P.rd_src_ptr);
    else if (k == "rd_src") from(P.rd_src);
    else if (k == "lev_ptr") from(P.lev_ptr);
    else if (k == "lev_fronts") from(P.lev_fronts);
    else if (k == "stats") v = { P.nF, P.nLevels, P.max_rows, (long long)P.nnzL, (long long)P.flops, (long long)P.pool_doubles };
    else { *out = nullptr; return -1; }
    *out = (long long *)malloc(sizeof(long long) * (v.size() + 1));
    if (!v.empty()) memcpy(*out, v.data(), sizeof(long long) * v.size());      // (an empty list has no data pointer: memcpy must not see it)
    return (long long)v.size();
}

}  // extern "C"

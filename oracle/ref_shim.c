/* ref_shim.c — TEST INFRASTRUCTURE (oracle side).  A few helpers compiled WITH the reference's own
 * headers and linked into oracle/_ref/libaprilsam_ref.so, so that Python (ctypes) can drive the
 * unmodified reference: the reference keeps zarray_add & friends `static inline`, and its solver
 * state (param->A, ->B, ->chol) is only reachable through its own struct definitions.
 *
 * Contains no reference code: it only CALLS the reference API.  Never linked into the product.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "aprilsam/aprilsam.h"

void rs_graph_add_node(april_graph_t *g, april_graph_node_t *n) { zarray_add(g->nodes, &n); }
void rs_graph_add_factor(april_graph_t *g, april_graph_factor_t *f) { zarray_add(g->factors, &f); }
int  rs_nnodes(april_graph_t *g) { return zarray_size(g->nodes); }
int  rs_nfactors(april_graph_t *g) { return zarray_size(g->factors); }

april_graph_node_t *rs_node(april_graph_t *g, int i)
{
    april_graph_node_t *n; zarray_get(g->nodes, i, &n); return n;
}
april_graph_factor_t *rs_factor(april_graph_t *g, int i)
{
    april_graph_factor_t *f; zarray_get(g->factors, i, &f); return f;
}

/* all states / l_points, 3 per node */
void rs_get_states(april_graph_t *g, double *out)
{
    for (int i = 0; i < zarray_size(g->nodes); i++) memcpy(out + 3*i, rs_node(g, i)->state, 24);
}
void rs_get_lpoints(april_graph_t *g, double *out)
{
    for (int i = 0; i < zarray_size(g->nodes); i++) memcpy(out + 3*i, rs_node(g, i)->l_point, 24);
}
void rs_set_states(april_graph_t *g, const double *in)
{
    for (int i = 0; i < zarray_size(g->nodes); i++) memcpy(rs_node(g, i)->state, in + 3*i, 24);
}

april_graph_cholesky_param_t *rs_param_create(void)
{
    april_graph_cholesky_param_t *p = calloc(1, sizeof(*p));
    april_graph_cholesky_param_init(p);
    return p;
}

/* factor->eval (at l_point for xyt, state for xytpos) or ->state_eval; copies out J0,J1,r,W,chi2 */
void rs_factor_eval(april_graph_t *g, int fidx, int use_state_eval, double *J0, double *J1, double *r, double *W, double *chi2)
{
    april_graph_factor_t *f = rs_factor(g, fidx);
    april_graph_factor_eval_t *e = (use_state_eval && f->state_eval) ? f->state_eval(f, g, NULL) : f->eval(f, g, NULL);
    memcpy(J0, e->jacobians[0]->data, 72);
    if (f->nnodes > 1) memcpy(J1, e->jacobians[1]->data, 72);
    memcpy(r, e->r, 24);
    memcpy(W, e->W->data, 72);
    *chi2 = e->chi2;
    april_graph_factor_eval_destroy(e);
}

/* upper-triangle entries of param->A (scalar coordinates of the reference's permuted system). */
int rs_param_A_nnz(april_graph_cholesky_param_t *p)
{
    smatd_t *A = p->A; int nz = 0;
    for (int i = 0; i < A->nrows; i++) nz += A->rows[i].nz;
    return nz;
}
void rs_param_A_dump(april_graph_cholesky_param_t *p, int *ri, int *ci, double *v)
{
    smatd_t *A = p->A; int k = 0;
    for (int i = 0; i < A->nrows; i++)
        for (int j = 0; j < A->rows[i].nz; j++) { ri[k] = i; ci[k] = A->rows[i].indices[j]; v[k] = A->rows[i].values[j]; k++; }
}
int rs_param_n(april_graph_cholesky_param_t *p) { return p->A ? p->A->nrows : 0; }
long long rs_param_U_nnz(april_graph_cholesky_param_t *p)
{
    smatd_t *u = p->chol->u; long long nz = 0;
    for (int i = 0; i < u->nrows; i++) nz += u->rows[i].nz;
    return nz;
}
double rs_param_U_sumsq(april_graph_cholesky_param_t *p)
{
    smatd_t *u = p->chol->u; double s = 0;
    for (int i = 0; i < u->nrows; i++) s += (double)u->rows[i].nz * u->rows[i].nz;
    return s;
}
void rs_param_B(april_graph_cholesky_param_t *p, double *out) { memcpy(out, p->B, sizeof(double) * p->A->nrows); }
void rs_param_ordering(april_graph_cholesky_param_t *p, int *out) { memcpy(out, p->ordering, sizeof(int) * p->nreordering); }
int  rs_tree_start_over(april_graph_cholesky_param_t *p) { return p->tr ? p->tr->start_over : -1; }
int  rs_tree_naffected(april_graph_cholesky_param_t *p) { return p->tr ? p->tr->naffected : -1; }

static double now_ms(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
/* time `iters` batch calls (the region examples/aprilsam_demo.c:103-107 times); ms per call out */
void rs_time_batch(april_graph_t *g, april_graph_cholesky_param_t *p, int iters, double *ms)
{
    for (int i = 0; i < iters; i++) {
        double t0 = now_ms();
        april_graph_cholesky(g, p);
        ms[i] = now_ms() - t0;
    }
}

/* bulk build: N xyt nodes (state=init=truth) and F factors; fb[i] < 0 => xytpos prior on fa[i]. */
void rs_build_from_arrays(april_graph_t *g, int N, const double *states, int F, const int *fa, const int *fb,
                          const double *z, const double *W)
{
    for (int i = 0; i < N; i++) {
        april_graph_node_t *n = april_graph_node_xyt_create(states + 3*i, states + 3*i, states + 3*i);
        zarray_add(g->nodes, &n);
    }
    matd_t *Wm = matd_create(3, 3);
    for (int i = 0; i < F; i++) {
        memcpy(Wm->data, W + 9*i, 72);
        double zz[3] = { z[3*i], z[3*i+1], z[3*i+2] };
        april_graph_factor_t *f = (fb[i] < 0) ? april_graph_factor_xytpos_create(fa[i], zz, NULL, Wm)
                                              : april_graph_factor_xyt_create(fa[i], fb[i], zz, NULL, Wm);
        zarray_add(g->factors, &f);
    }
    matd_destroy(Wm);
}

/* search tree introspection (aprilsam.h:190-219) for the tests of the bookkeeping model */
int  rs_tree_nnodes(april_graph_cholesky_param_t *p) { return p->tr ? p->tr->nnodes : 0; }
void rs_tree_parents(april_graph_cholesky_param_t *p, int *out) { for (int i = 0; i < p->tr->nnodes; i++) out[i] = p->tr->nodes[i].parent; }
void rs_tree_labels(april_graph_cholesky_param_t *p, int *changed, int *relin)
{
    for (int i = 0; i < p->tr->nnodes; i++) { changed[i] = p->tr->nodes[i].label_changed; relin[i] = p->tr->nodes[i].label_relinearized; }
}
int  rs_tree_root(april_graph_cholesky_param_t *p) { return (int)(p->tr->root - p->tr->nodes); }

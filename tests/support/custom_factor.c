/* Test infrastructure: two factor types NEITHER library knows (type tags 77 and 78), implemented against the public
 * object layout (include/aprilsam_amd.h PART 1 == aprilsam/aprilsam.h:75-146) with their own eval / state_eval / copy /
 * destroy function pointers.  The reference evaluates them through the vtable (aprilsam.c:156); this build must do the
 * same on its host-fallback path.  Objects are allocated with calloc so that either
 * library's april_graph_factor_eval_destroy / destroy paths can free them.
 *
 *   type 77  "xy"       two poses, 2 residuals: position of b in a's frame, r = z - h(a, b), W 2x2
 *   type 78  "heading"  one pose, 1 residual: r = wrap(z - theta), W 1x1
 *   type 79  "midpoint" THREE poses a, b, c, 2 residuals: b seen from the midpoint of a and c, in a's heading,
 *                       r = z - R(theta_a)^T (p_b - (p_a + p_c) / 2), W 2x2 -- exercises factor->nnodes > 2, which the
 *                       reference's assembly loops handle generically (aprilsam.c:159-192)
 * All linearise at the nodes' l_point in eval() and at state in state_eval().
 *
 *   gcc -O2 -fPIC -shared -Iinclude tests/support/custom_factor.c -o <out>.so -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "aprilsam_amd.h"

static matd_t *matd_new(unsigned r, unsigned c) {
    matd_t *m = (matd_t *)calloc(1, sizeof(matd_t) + sizeof(double) * r * c);
    m->nrows = r; m->ncols = c;
    return m;
}
static double wrap_pi(double v) {            /* [-pi, pi), same convention as the reference's mod2pi */
    const double twopi = 6.2831853071795862319959, pi = 3.141592653589793238462643383279502884196;
    double vin = v + pi;
    return (vin - twopi * floor(vin / twopi)) - pi;
}
static april_graph_node_t *node_of(april_graph_t *g, int i) { return ((april_graph_node_t **)g->nodes->data)[i]; }

/* ---------------------------------------------------------------- type 77 */
static april_graph_factor_eval_t *xy_eval_at(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e, int at_state) {
    if (!e) {
        e = (april_graph_factor_eval_t *)calloc(1, sizeof(*e));
        e->jacobians = (matd_t **)calloc(3, sizeof(matd_t *));
        e->jacobians[0] = matd_new(2, 3); e->jacobians[1] = matd_new(2, 3);
        e->r = (double *)calloc(2, sizeof(double));
        e->W = matd_new(2, 2);
    }
    e->length = 2;
    const double *pa = at_state ? node_of(g, f->nodes[0])->state : node_of(g, f->nodes[0])->l_point;
    const double *pb = at_state ? node_of(g, f->nodes[1])->state : node_of(g, f->nodes[1])->l_point;
    const double c = cos(pa[2]), s = sin(pa[2]), dx = pb[0] - pa[0], dy = pb[1] - pa[1];
    const double hx = c * dx + s * dy, hy = -s * dx + c * dy;
    double *Ja = e->jacobians[0]->data, *Jb = e->jacobians[1]->data;
    Ja[0] = -c; Ja[1] = -s; Ja[2] = -s * dx + c * dy;
    Ja[3] = s;  Ja[4] = -c; Ja[5] = -c * dx - s * dy;
    Jb[0] = c;  Jb[1] = s;  Jb[2] = 0;
    Jb[3] = -s; Jb[4] = c;  Jb[5] = 0;
    e->r[0] = f->u.common.z[0] - hx; e->r[1] = f->u.common.z[1] - hy;
    memcpy(e->W->data, f->u.common.W->data, 4 * sizeof(double));
    const double *W = e->W->data;
    e->chi2 = e->r[0] * (W[0] * e->r[0] + W[1] * e->r[1]) + e->r[1] * (W[2] * e->r[0] + W[3] * e->r[1]);
    return e;
}
static april_graph_factor_eval_t *xy_eval(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e) { return xy_eval_at(f, g, e, 0); }
static april_graph_factor_eval_t *xy_state_eval(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e) { return xy_eval_at(f, g, e, 1); }
static void common_destroy(april_graph_factor_t *f) {
    free(f->nodes); free(f->u.common.z); free(f->u.common.ztruth); free(f->u.common.W); free(f);
}
april_graph_factor_t *custom_xy_create(int a, int b, const double *z, const double *W4);
static april_graph_factor_t *xy_copy(april_graph_factor_t *f) { return custom_xy_create(f->nodes[0], f->nodes[1], f->u.common.z, f->u.common.W->data); }
april_graph_factor_t *custom_xy_create(int a, int b, const double *z, const double *W4) {
    april_graph_factor_t *f = (april_graph_factor_t *)calloc(1, sizeof(*f));
    f->type = 77; f->nnodes = 2; f->length = 2;
    f->nodes = (int *)calloc(2, sizeof(int)); f->nodes[0] = a; f->nodes[1] = b;
    f->copy = xy_copy; f->eval = xy_eval; f->state_eval = xy_state_eval; f->destroy = common_destroy;
    f->u.common.z = (double *)calloc(2, sizeof(double)); memcpy(f->u.common.z, z, 2 * sizeof(double));
    f->u.common.W = matd_new(2, 2); memcpy(f->u.common.W->data, W4, 4 * sizeof(double));
    return f;
}

/* ---------------------------------------------------------------- type 78 */
static april_graph_factor_eval_t *hd_eval_at(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e, int at_state) {
    if (!e) {
        e = (april_graph_factor_eval_t *)calloc(1, sizeof(*e));
        e->jacobians = (matd_t **)calloc(2, sizeof(matd_t *));
        e->jacobians[0] = matd_new(1, 3);
        e->r = (double *)calloc(1, sizeof(double));
        e->W = matd_new(1, 1);
    }
    e->length = 1;
    const double *pa = at_state ? node_of(g, f->nodes[0])->state : node_of(g, f->nodes[0])->l_point;
    e->jacobians[0]->data[0] = 0; e->jacobians[0]->data[1] = 0; e->jacobians[0]->data[2] = 1;
    e->r[0] = wrap_pi(f->u.common.z[0] - pa[2]);
    e->W->data[0] = f->u.common.W->data[0];
    e->chi2 = e->r[0] * e->W->data[0] * e->r[0];
    return e;
}
static april_graph_factor_eval_t *hd_eval(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e) { return hd_eval_at(f, g, e, 0); }
static april_graph_factor_eval_t *hd_state_eval(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e) { return hd_eval_at(f, g, e, 1); }
april_graph_factor_t *custom_heading_create(int a, double z, double w);
static april_graph_factor_t *hd_copy(april_graph_factor_t *f) { return custom_heading_create(f->nodes[0], f->u.common.z[0], f->u.common.W->data[0]); }
april_graph_factor_t *custom_heading_create(int a, double z, double w) {
    april_graph_factor_t *f = (april_graph_factor_t *)calloc(1, sizeof(*f));
    f->type = 78; f->nnodes = 1; f->length = 1;
    f->nodes = (int *)calloc(1, sizeof(int)); f->nodes[0] = a;
    f->copy = hd_copy; f->eval = hd_eval; f->state_eval = hd_state_eval; f->destroy = common_destroy;
    f->u.common.z = (double *)calloc(1, sizeof(double)); f->u.common.z[0] = z;
    f->u.common.W = matd_new(1, 1); f->u.common.W->data[0] = w;
    return f;
}

/* ---------------------------------------------------------------- type 79 */
static april_graph_factor_eval_t *mid_eval_at(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e, int at_state) {
    if (!e) {
        e = (april_graph_factor_eval_t *)calloc(1, sizeof(*e));
        e->jacobians = (matd_t **)calloc(4, sizeof(matd_t *));
        for (int k = 0; k < 3; k++) e->jacobians[k] = matd_new(2, 3);
        e->r = (double *)calloc(2, sizeof(double));
        e->W = matd_new(2, 2);
    }
    e->length = 2;
    const double *p[3];
    for (int k = 0; k < 3; k++) p[k] = at_state ? node_of(g, f->nodes[k])->state : node_of(g, f->nodes[k])->l_point;
    const double c = cos(p[0][2]), s = sin(p[0][2]);
    const double d0 = p[1][0] - 0.5 * (p[0][0] + p[2][0]), d1 = p[1][1] - 0.5 * (p[0][1] + p[2][1]);
    const double h0 = c * d0 + s * d1, h1 = -s * d0 + c * d1;
    double *Ja = e->jacobians[0]->data, *Jb = e->jacobians[1]->data, *Jc = e->jacobians[2]->data;
    Ja[0] = -0.5 * c; Ja[1] = -0.5 * s; Ja[2] = h1;
    Ja[3] = 0.5 * s;  Ja[4] = -0.5 * c; Ja[5] = -h0;
    Jb[0] = c;  Jb[1] = s; Jb[2] = 0;
    Jb[3] = -s; Jb[4] = c; Jb[5] = 0;
    Jc[0] = -0.5 * c; Jc[1] = -0.5 * s; Jc[2] = 0;
    Jc[3] = 0.5 * s;  Jc[4] = -0.5 * c; Jc[5] = 0;
    e->r[0] = f->u.common.z[0] - h0; e->r[1] = f->u.common.z[1] - h1;
    memcpy(e->W->data, f->u.common.W->data, 4 * sizeof(double));
    const double *W = e->W->data;
    e->chi2 = e->r[0] * (W[0] * e->r[0] + W[1] * e->r[1]) + e->r[1] * (W[2] * e->r[0] + W[3] * e->r[1]);
    return e;
}
static april_graph_factor_eval_t *mid_eval(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e) { return mid_eval_at(f, g, e, 0); }
static april_graph_factor_eval_t *mid_state_eval(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e) { return mid_eval_at(f, g, e, 1); }
april_graph_factor_t *custom_midpoint_create(int a, int b, int c, const double *z, const double *W4);
static april_graph_factor_t *mid_copy(april_graph_factor_t *f) { return custom_midpoint_create(f->nodes[0], f->nodes[1], f->nodes[2], f->u.common.z, f->u.common.W->data); }
april_graph_factor_t *custom_midpoint_create(int a, int b, int c, const double *z, const double *W4) {
    april_graph_factor_t *f = (april_graph_factor_t *)calloc(1, sizeof(*f));
    f->type = 79; f->nnodes = 3; f->length = 2;
    f->nodes = (int *)calloc(3, sizeof(int)); f->nodes[0] = a; f->nodes[1] = b; f->nodes[2] = c;
    f->copy = mid_copy; f->eval = mid_eval; f->state_eval = mid_state_eval; f->destroy = common_destroy;
    f->u.common.z = (double *)calloc(2, sizeof(double)); memcpy(f->u.common.z, z, 2 * sizeof(double));
    f->u.common.W = matd_new(2, 2); memcpy(f->u.common.W->data, W4, 4 * sizeof(double));
    return f;
}

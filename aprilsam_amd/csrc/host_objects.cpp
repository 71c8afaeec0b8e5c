// host_objects.cpp — host-side graph objects with the reference's ABI (include/aprilsam_amd.h PART 1/3).
//
// These constructors let a caller build a graph without libaprilsam: objects are laid out exactly like
// the reference's (april_graph.c:329-364, april_graph_xyt.c:276-298,420-438, april_graph_xytpos.c:191-211)
// and carry working vtable entries (copy / eval / state_eval / destroy, update / relinearize / destroy)
// because reference-compiled host code calls them directly (examples/aprilsam_demo.c:183 nb->relinearize,
// :166 exist_factor->copy, april_graph.c:335-346 ->destroy).  The vtable entries are a courtesy to such
// callers; the solver entry points never call them — factors of type 1/2 are evaluated by the HIP
// kernels (csrc/kernels.hip.h).
#include <cmath>
#include <string>
#include <utility>
#include <vector>
#include <cstdlib>
#include <cstring>

#include "../../include/aprilsam_amd.h"
#include "solver.h"

namespace {

double *dup3(const double *v) {              // doubles_dup (doubles_floats_impl.h:68), NULL-tolerant
    if (!v) return nullptr;
    double *r = (double *)malloc(3 * sizeof(double));
    memcpy(r, v, 3 * sizeof(double));
    return r;
}
matd_t *matd33(const double *data) {
    matd_t *m = (matd_t *)calloc(1, sizeof(matd_t) + 9 * sizeof(double));
    m->nrows = 3; m->ncols = 3;
    if (data) memcpy(m->data, data, 9 * sizeof(double));
    return m;
}
double mod2pi_h(double v) {                   // math_util.h:113-122
    const double TWOPI = 6.2831853071795862319959, PI_ = 3.141592653589793238462643383279502884196;
    double vin = v + PI_;
    return (vin - TWOPI * floor(vin / TWOPI)) - PI_;
}
void zarray_append(zarray_t *za, const void *p) {      // zarray.h:152-189 semantics (doubling growth)
    if (za->size + 1 > za->alloc) {
        int na = za->alloc;
        while (na < za->size + 1) { na *= 2; if (na < 8) na = 8; }
        za->data = (char *)realloc(za->data, (size_t)na * za->el_sz);
        za->alloc = na;
    }
    memcpy(za->data + (size_t)za->size * za->el_sz, p, za->el_sz);
    za->size++;
}
zarray_t *zarray_new(size_t el_sz) {
    zarray_t *za = (zarray_t *)calloc(1, sizeof(zarray_t));
    za->el_sz = el_sz;
    return za;
}

april_graph_factor_eval_t *eval_alloc(int njac) {
    april_graph_factor_eval_t *e = (april_graph_factor_eval_t *)calloc(1, sizeof(*e));
    e->jacobians = (matd_t **)calloc(njac + 1, sizeof(matd_t *));      // NULL-terminated
    for (int i = 0; i < njac; i++) e->jacobians[i] = matd33(nullptr);
    e->r = (double *)calloc(3, sizeof(double));
    e->W = matd33(nullptr);
    return e;
}
void eval_finish(april_graph_factor_t *f, april_graph_factor_eval_t *e) {
    e->length = 3;
    memcpy(e->W->data, f->u.common.W->data, 72);
    const double *w = e->W->data, *r = e->r;
    double X0 = w[0] * r[0] + w[1] * r[1] + w[2] * r[2];
    double X1 = w[3] * r[0] + w[4] * r[1] + w[5] * r[2];
    double X2 = w[6] * r[0] + w[7] * r[1] + w[8] * r[2];
    e->chi2 = r[0] * X0 + r[1] * X1 + r[2] * X2;
}

// ---- xyt factor ------------------------------------------------------------------------------------------
april_graph_factor_eval_t *xyt_eval_at(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e, bool at_state) {
    if (!e) e = eval_alloc(2);
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    const double *pa = at_state ? ns[f->nodes[0]]->state : ns[f->nodes[0]]->l_point;
    const double *pb = at_state ? ns[f->nodes[1]]->state : ns[f->nodes[1]]->l_point;
    double ca = cos(pa[2]), sa = sin(pa[2]);
    double dx = pb[0] - pa[0], dy = pb[1] - pa[1];
    double zh0 = ca * dx + sa * dy, zh1 = -sa * dx + ca * dy, zh2 = pb[2] - pa[2];
    const double J0[9] = { -ca, -sa, -sa * dx + ca * dy, sa, -ca, -ca * dx - sa * dy, 0, 0, -1 };
    const double J1[9] = { ca, sa, 0, -sa, ca, 0, 0, 0, 1 };
    memcpy(e->jacobians[0]->data, J0, 72);
    memcpy(e->jacobians[1]->data, J1, 72);
    const double *z = f->u.common.z;
    e->r[0] = z[0] - zh0; e->r[1] = z[1] - zh1; e->r[2] = mod2pi_h(z[2] - zh2);
    eval_finish(f, e);
    return e;
}
april_graph_factor_eval_t *xyt_eval(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e) { return xyt_eval_at(f, g, e, false); }
april_graph_factor_eval_t *xyt_state_eval(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e) { return xyt_eval_at(f, g, e, true); }
// ---- attributes ----------------------------------------------------------------------------------------------
// The reference hangs a zhash of typed values off node->attr / factor->attr / graph->attr (april_graph.c:101-176).  This
// library keeps only what the `.graph` files of the path need: string values ("type" = "odom" / "scan" on the demo's
// factors, examples/aprilsam_demo.c:84-86), in insertion order.  The block starts with a tag so that an attr pointer
// set by the REFERENCE library (whose block starts with a heap pointer) is recognised as foreign and left alone.
constexpr unsigned long long ATTR_TAG = 0x617474725f616d64ULL;       // "attr_amd"
struct AmdAttr { unsigned long long tag = ATTR_TAG; std::vector<std::pair<std::string, std::string>> kv; };
AmdAttr *own_attr(const void *p) { return (p && *(const unsigned long long *)p == ATTR_TAG) ? (AmdAttr *)p : nullptr; }
void attr_free(void *p) { if (AmdAttr *a = own_attr(p)) delete a; }
void *attr_clone(const void *p) { const AmdAttr *a = own_attr(p); return a ? new AmdAttr(*a) : nullptr; }
int attr_put(void **slot, const char *key, const char *value) {
    if (!key || !value) return -1;
    if (*slot && !own_attr(*slot)) return -2;                        // attributes owned by another library
    if (!*slot) *slot = new AmdAttr();
    AmdAttr *a = (AmdAttr *)*slot;
    for (auto &kv : a->kv) if (kv.first == key) { kv.second = value; return 0; }
    a->kv.emplace_back(key, value);
    return 0;
}
const char *attr_get(const void *p, const char *key) {
    const AmdAttr *a = own_attr(p);
    if (!a || !key) return nullptr;
    for (auto &kv : a->kv) if (kv.first == key) return kv.second.c_str();
    return nullptr;
}

void factor_destroy(april_graph_factor_t *f) {
    free(f->nodes); free(f->u.common.z); free(f->u.common.ztruth); free(f->u.common.W);
    attr_free(f->attr);
    free(f);
}
april_graph_factor_t *xyt_copy(april_graph_factor_t *f) {
    april_graph_factor_t *c = april_graph_factor_xyt_create(f->nodes[0], f->nodes[1], f->u.common.z, f->u.common.ztruth, f->u.common.W);
    c->attr = attr_clone(f->attr);            // the reference's copy keeps the attributes (the demo reads "type" from the copy)
    return c;
}

// ---- xytpos factor ----------------------------------------------------------------------------------------
april_graph_factor_eval_t *xytpos_eval(april_graph_factor_t *f, april_graph_t *g, april_graph_factor_eval_t *e) {
    if (!e) e = eval_alloc(1);
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    const double *pa = ns[f->nodes[0]]->state;                    // state, not l_point (april_graph_xytpos.c:83-85)
    double *J = e->jacobians[0]->data;
    J[0] = 1; J[4] = 1; J[8] = 1;
    const double *z = f->u.common.z;
    e->r[0] = z[0] - pa[0]; e->r[1] = z[1] - pa[1]; e->r[2] = mod2pi_h(z[2] - pa[2]);
    eval_finish(f, e);
    return e;
}
april_graph_factor_t *xytpos_copy(april_graph_factor_t *f) {
    april_graph_factor_t *c = april_graph_factor_xytpos_create(f->nodes[0], f->u.common.z, f->u.common.ztruth, f->u.common.W);
    c->attr = attr_clone(f->attr);
    return c;
}

// ---- xyt node ----------------------------------------------------------------------------------------------
void node_update(april_graph_node_t *n, double *d) {             // april_graph_xyt.c:302-314
    for (int i = 0; i < 3; i++) if (std::isnan(d[i])) return;
    for (int i = 0; i < 3; i++) n->state[i] = n->l_point[i] + d[i];
    for (int i = 0; i < 3; i++) n->delta_X[i] = d[i];
    n->state[2] = mod2pi_h(n->state[2]);
}
void node_relinearize(april_graph_node_t *n) { memcpy(n->l_point, n->state, 24); }    // april_graph_xyt.c:316-320
void node_destroy(april_graph_node_t *n) {
    free(n->state); free(n->init); free(n->truth); free(n->l_point); free(n->delta_X); attr_free(n->attr); free(n);
}
april_graph_node_t *node_copy(april_graph_node_t *n) {
    april_graph_node_t *c = april_graph_node_xyt_create(n->state, n->init, n->truth);
    memcpy(c->l_point, n->l_point, 24); memcpy(c->delta_X, n->delta_X, 24);
    c->attr = attr_clone(n->attr);
    return c;
}

}  // namespace

extern "C" {

april_graph_t *april_graph_create(void) {
    april_graph_t *g = (april_graph_t *)calloc(1, sizeof(april_graph_t));
    g->factors = zarray_new(sizeof(april_graph_factor_t *));
    g->nodes = zarray_new(sizeof(april_graph_node_t *));
    return g;
}

void april_graph_destroy(april_graph_t *g) {
    if (!g) return;
    asam::drop_graph_pack(g);
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    for (int i = 0; i < g->nodes->size; i++) ns[i]->destroy(ns[i]);
    april_graph_factor_t **fs = (april_graph_factor_t **)g->factors->data;
    for (int i = 0; i < g->factors->size; i++) fs[i]->destroy(fs[i]);
    free(g->nodes->data); free(g->nodes);
    free(g->factors->data); free(g->factors);
    attr_free(g->attr);
    free(g);
}

april_graph_node_t *april_graph_node_xyt_create(const double *state, const double *init, const double *truth) {
    april_graph_node_t *n = (april_graph_node_t *)calloc(1, sizeof(april_graph_node_t));
    n->type = APRIL_GRAPH_NODE_XYT_TYPE;
    n->length = 3;
    n->state = dup3(state); n->init = dup3(init); n->truth = dup3(truth);
    n->l_point = dup3(state);
    n->delta_X = (double *)calloc(3, sizeof(double));
    n->update = node_update; n->copy = node_copy; n->relinearize = node_relinearize; n->destroy = node_destroy;
    return n;
}

april_graph_factor_t *april_graph_factor_xyt_create(int a, int b, const double *z, const double *ztruth, const matd_t *W) {
    april_graph_factor_t *f = (april_graph_factor_t *)calloc(1, sizeof(april_graph_factor_t));
    f->type = APRIL_GRAPH_FACTOR_XYT_TYPE;
    f->nnodes = 2;
    f->nodes = (int *)calloc(2, sizeof(int));
    f->nodes[0] = a; f->nodes[1] = b;
    f->length = 3;
    f->copy = xyt_copy; f->eval = xyt_eval; f->state_eval = xyt_state_eval; f->destroy = factor_destroy;
    f->u.common.z = dup3(z); f->u.common.ztruth = dup3(ztruth); f->u.common.W = matd33(W ? W->data : nullptr);
    return f;
}

april_graph_factor_t *april_graph_factor_xytpos_create(int a, double *z, double *ztruth, matd_t *W) {
    april_graph_factor_t *f = (april_graph_factor_t *)calloc(1, sizeof(april_graph_factor_t));
    f->type = APRIL_GRAPH_FACTOR_XYTPOS_TYPE;
    f->nnodes = 1;
    f->nodes = (int *)calloc(1, sizeof(int));
    f->nodes[0] = a;
    f->length = 3;
    f->copy = xytpos_copy; f->eval = xytpos_eval; f->destroy = factor_destroy;
    f->u.common.z = dup3(z); f->u.common.ztruth = dup3(ztruth); f->u.common.W = matd33(W ? W->data : nullptr);
    return f;
}

void april_graph_factor_eval_destroy(april_graph_factor_eval_t *e) {      // april_graph.c:33-49
    if (!e) return;
    for (int i = 0; i < e->length; i++) { if (!e->jacobians[i]) break; free(e->jacobians[i]); }
    free(e->jacobians); free(e->r); free(e->W); free(e);
}

int april_graph_dof(april_graph_t *g) {                                  // april_graph.c:57-77
    int fd = 0, sd = 0;
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    for (int i = 0; i < g->nodes->size; i++) sd += ns[i]->length;
    april_graph_factor_t **fs = (april_graph_factor_t **)g->factors->data;
    for (int i = 0; i < g->factors->size; i++) fd += fs[i]->length;
    return fd - sd;
}

void aprilsam_amd_graph_add_node(april_graph_t *g, april_graph_node_t *n) { zarray_append(g->nodes, &n); }
void aprilsam_amd_graph_add_factor(april_graph_t *g, april_graph_factor_t *f) { zarray_append(g->factors, &f); }

int aprilsam_amd_attr_put_string(void **attr_slot, const char *key, const char *value) { return attr_slot ? attr_put(attr_slot, key, value) : -1; }
const char *aprilsam_amd_attr_get_string(const void *attr, const char *key) { return attr_get(attr, key); }
int aprilsam_amd_attr_count(const void *attr) { const AmdAttr *a = own_attr(attr); return a ? (int)a->kv.size() : 0; }
int aprilsam_amd_attr_item(const void *attr, int i, const char **key, const char **value) {
    const AmdAttr *a = own_attr(attr);
    if (!a || i < 0 || i >= (int)a->kv.size()) return -1;
    *key = a->kv[i].first.c_str(); *value = a->kv[i].second.c_str();
    return 0;
}

void aprilsam_amd_graph_from_arrays(april_graph_t *g, int N, const double *states, int F, const int *fa, const int *fb,
                                    const double *z, const double *W) {
    for (int i = 0; i < N; i++) {
        april_graph_node_t *n = april_graph_node_xyt_create(states + 3 * i, states + 3 * i, states + 3 * i);
        zarray_append(g->nodes, &n);
    }
    matd_t *Wm = matd33(nullptr);
    for (int i = 0; i < F; i++) {
        memcpy(Wm->data, W + 9 * (size_t)i, 72);
        double zz[3] = { z[3 * (size_t)i], z[3 * (size_t)i + 1], z[3 * (size_t)i + 2] };
        april_graph_factor_t *f = fb[i] < 0 ? april_graph_factor_xytpos_create(fa[i], zz, nullptr, Wm)
                                            : april_graph_factor_xyt_create(fa[i], fb[i], zz, nullptr, Wm);
        zarray_append(g->factors, &f);
    }
    free(Wm);
}

// bulk read of the node objects (state / l_point / delta_X, april_graph_xyt.c:302-314 fields): what a checker of a big graph
// needs without a million ctypes round trips; any of the three destinations may be null
void aprilsam_amd_graph_node_arrays(const april_graph_t *g, double *state, double *l_point, double *delta_X) {
    const int N = g && g->nodes ? g->nodes->size : 0;
    april_graph_node_t *const *ns = N ? (april_graph_node_t *const *)g->nodes->data : nullptr;
    for (int i = 0; i < N; i++) {
        const april_graph_node_t *n = ns[i];
        if (state) memcpy(state + 3 * (size_t)i, n->state, 24);
        if (l_point) memcpy(l_point + 3 * (size_t)i, n->l_point, 24);
        if (delta_X) memcpy(delta_X + 3 * (size_t)i, n->delta_X, 24);
    }
}

// ---- synthetic Manhattan lattice, SURVEY.md §8(d) config 4/5 -------------------------------------------
// K x K poses on a unit grid, snake numbering id(r,c) = r*K + (r odd ? K-1-c : c); truth = (c, r, r odd ? pi : 0);
// splitmix64 PRNG (state 0x9E3779B97F4A7C15, first output after state += gamma); u01 = (x >> 11) * 2^-53;
// gauss = sqrt(-2 ln u1) cos(2 pi u2), u1 clamped to >= 1e-300, u1 drawn first.
namespace {
struct SplitMix {
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    unsigned long long next() {
        s += 0x9E3779B97F4A7C15ull;
        unsigned long long z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double u01() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double gauss() {
        double u1 = u01(), u2 = u01();
        if (u1 < 1e-300) u1 = 1e-300;
        return sqrt(-2.0 * log(u1)) * cos(2.0 * 3.141592653589793238462643383279502884196 * u2);
    }
};
}  // namespace

int aprilsam_amd_lattice_arrays(int K, double *states, int *fa, int *fb, double *z, double *W) {
    const double PI_ = 3.141592653589793238462643383279502884196;
    auto id = [&](int r, int c) { return r * K + ((r & 1) ? K - 1 - c : c); };
    const int N = K * K;
    double *truth = (double *)malloc(sizeof(double) * 3 * (size_t)N);
    for (int r = 0; r < K; r++)
        for (int c = 0; c < K; c++) { int i = id(r, c); truth[3 * i] = c; truth[3 * i + 1] = r; truth[3 * i + 2] = (r & 1) ? PI_ : 0.0; }
    SplitMix rng;
    for (int i = 0; i < N; i++) {                 // id ascending; node 0 draws, then is pinned to the origin
        double g0 = rng.gauss(), g1 = rng.gauss(), g2 = rng.gauss();
        states[3 * i] = truth[3 * i] + 0.2 * g0; states[3 * i + 1] = truth[3 * i + 1] + 0.2 * g1; states[3 * i + 2] = truth[3 * i + 2] + 0.05 * g2;
    }
    states[0] = states[1] = states[2] = 0;
    int F = 0;
    const int d[4][2] = { { 0, 1 }, { 1, 0 }, { 1, 1 }, { 1, -1 } };
    for (int r = 0; r < K; r++)
        for (int c = 0; c < K; c++)
            for (int k = 0; k < 4; k++) {
                int r2 = r + d[k][0], c2 = c + d[k][1];
                if (r2 < 0 || r2 >= K || c2 < 0 || c2 >= K) continue;
                int a = id(r, c), b = id(r2, c2);
                if (a > b) { int t = a; a = b; b = t; }
                const double *ta = truth + 3 * a, *tb = truth + 3 * b;
                double ca = cos(ta[2]), sa = sin(ta[2]), dx = tb[0] - ta[0], dy = tb[1] - ta[1];
                double g0 = rng.gauss(), g1 = rng.gauss(), g2 = rng.gauss();
                fa[F] = a; fb[F] = b;
                z[3 * (size_t)F] = (ca * dx + sa * dy) + 0.05 * g0;           // truth_a^-1 o truth_b (doubles_floats_impl.h:619)
                z[3 * (size_t)F + 1] = (-sa * dx + ca * dy) + 0.05 * g1;
                z[3 * (size_t)F + 2] = mod2pi_h((tb[2] - ta[2]) + 0.01 * g2);
                double *w = W + 9 * (size_t)F;
                memset(w, 0, 72); w[0] = 400; w[4] = 400; w[8] = 1e4;
                F++;
            }
    fa[F] = 0; fb[F] = -1;                        // prior on node 0, W = diag(1e4, 1e4, 1e3), z = 0
    z[3 * (size_t)F] = z[3 * (size_t)F + 1] = z[3 * (size_t)F + 2] = 0;
    { double *w = W + 9 * (size_t)F; memset(w, 0, 72); w[0] = 1e4; w[4] = 1e4; w[8] = 1e3; }
    F++;
    free(truth);
    return F;
}

int aprilsam_amd_make_lattice(april_graph_t *g, int K) {
    const size_t N = (size_t)K * K, Fmax = 4 * N + 1;
    double *st = (double *)malloc(24 * N), *z = (double *)malloc(24 * Fmax), *W = (double *)malloc(72 * Fmax);
    int *fa = (int *)malloc(4 * Fmax), *fb = (int *)malloc(4 * Fmax);
    int F = aprilsam_amd_lattice_arrays(K, st, fa, fb, z, W);
    aprilsam_amd_graph_from_arrays(g, (int)N, st, F, fa, fb, z, W);
    free(st); free(z); free(W); free(fa); free(fb);
    return F;
}

}  // extern "C"

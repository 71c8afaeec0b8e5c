/* oracle.c — plain-C CPU restatement of AprilSAM's batch Gauss-Newton step (reference
 * xipengwang/AprilSAM @ v1).  TEST INFRASTRUCTURE — the checker the HIP path is compared against; it is
 * never linked into, loaded by, or shipped with the product library (aprilsam_amd/).
 *
 * Pinning: tests/test_oracle_vs_golden.py checks every function here against golden vectors produced by
 * the UNMODIFIED reference compiled from /root/reference (oracle/gen_golden.py, fixtures under
 * tests/golden/): per-factor J/r/chi2, the M3500 10-iteration chi^2 sequence + final states, the
 * 6-pose tutorial, and small lattices.  When oracle/_ref/libaprilsam_ref.so is present the same tests
 * also compare against the reference live.
 *
 * What is restated, with the reference lines followed:
 *   orc_mod2pi            common/math_util.h:113-122
 *   orc_factor_eval       april_graph_xyt.c:62-124 (xyt), april_graph_xytpos.c:63-102 (xytpos)
 *   orc_chi2              april_graph.c:79-98
 *   normal equations      aprilsam.c:141-204 ((J'W)J association of matd_op, upper triangle only,
 *                         Tikhonov on the diagonal)
 *   ordering              node-level greedy minimum degree with explicit clique formation, the mechanism
 *                         of aprilsam.c:1148-1199 without its "recent poses last" constraint (results are
 *                         ordering independent to ~1e-10) -- EXCEPT when a factor's W is not
 *                         symmetric as given: the upper-triangle rule of aprilsam.c:171 then makes the normal
 *                         equations depend on the order, and the *_ordered entry points take the reference's
 *                         own order (param->ordering) as data (tests/golden/asym_*.npz)
 *   Cholesky              up-looking sparse Cholesky = the published CSparse algorithm the reference
 *                         calls at aprilsam.c:233-234 (cs_schol/cs_chol, csparse.c:462-512: elimination
 *                         tree, row reach, sparse triangular solve per row), restated from T. Davis,
 *                         "Direct Methods for Sparse Linear Systems", ch. 4
 *   solves                U'y = B then Ux = y (smatd.c:1051-1114)
 *   update                state = l_point + dx, theta wrapped, NaN guard (april_graph_xyt.c:302-314)
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_TWOPI 6.2831853071795862319959
#define ORC_PI 3.141592653589793238462643383279502884196

double orc_mod2pi(double v)
{
    double vin = v + ORC_PI;
    double pos = vin - ORC_TWOPI * floor(vin / ORC_TWOPI);
    return pos - ORC_PI;
}

double orc_factor_eval(int binary, const double *pa, const double *pb, const double *z, const double *W,
                       double *J0, double *J1, double *r)
{
    if (binary) {
        double xa = pa[0], ya = pa[1], ta = pa[2];
        double xb = pb[0], yb = pb[1], tb = pb[2];
        double ca = cos(ta), sa = sin(ta);
        double dx = xb - xa, dy = yb - ya;
        double zhat0 = ca * dx + sa * dy;
        double zhat1 = -sa * dx + ca * dy;
        double zhat2 = tb - ta;
        J0[0] = -ca; J0[1] = -sa; J0[2] = -sa * dx + ca * dy;
        J0[3] = sa;  J0[4] = -ca; J0[5] = -ca * dx - sa * dy;
        J0[6] = 0;   J0[7] = 0;   J0[8] = -1;
        J1[0] = ca;  J1[1] = sa;  J1[2] = 0;
        J1[3] = -sa; J1[4] = ca;  J1[5] = 0;
        J1[6] = 0;   J1[7] = 0;   J1[8] = 1;
        r[0] = z[0] - zhat0;
        r[1] = z[1] - zhat1;
        r[2] = orc_mod2pi(z[2] - zhat2);
    } else {
        memset(J0, 0, 72);
        J0[0] = J0[4] = J0[8] = 1;
        r[0] = z[0] - pa[0];
        r[1] = z[1] - pa[1];
        r[2] = orc_mod2pi(z[2] - pa[2]);
    }
    double X[3];
    for (int i = 0; i < 3; i++) X[i] = W[3 * i] * r[0] + W[3 * i + 1] * r[1] + W[3 * i + 2] * r[2];
    return r[0] * X[0] + r[1] * X[1] + r[2] * X[2];
}

double orc_chi2(int N, const double *st, int F, const int *fa, const int *fb, const double *z, const double *W)
{
    (void)N;
    double chi2 = 0, J0[9], J1[9], r[3];
    for (int f = 0; f < F; f++) {
        int bin = fb[f] >= 0;
        double c = orc_factor_eval(bin, st + 3 * fa[f], bin ? st + 3 * fb[f] : NULL, z + 3 * f, W + 9 * f, J0, J1, r);
        chi2 += bin ? 0.5 * c : c;
    }
    return chi2;
}

/* c = a' * b and c = a * b for row-major 3x3, accumulating k = 0,1,2 in order (matd.c:241-247) */
static void mul_atb(const double *a, const double *b, double *c)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += a[3 * k + i] * b[3 * k + j];
            c[3 * i + j] = acc;
        }
}
static void mul_ab(const double *a, const double *b, double *c)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += a[3 * i + k] * b[3 * k + j];
            c[3 * i + j] = acc;
        }
}

/* ---- triplet accumulation of the upper triangle --------------------------------------------------- */
typedef struct { int row, col; double v; long seq; } trip_t;
typedef struct { trip_t *t; long n, cap; } trips_t;
static void trip_add(trips_t *T, int row, int col, double v)
{
    if (T->n == T->cap) { T->cap = T->cap ? 2 * T->cap : 1024; T->t = realloc(T->t, sizeof(trip_t) * T->cap); }
    T->t[T->n].row = row; T->t[T->n].col = col; T->t[T->n].v = v; T->t[T->n].seq = T->n; T->n++;
}
static int trip_cmp(const void *a, const void *b)
{
    const trip_t *x = a, *y = b;
    if (x->col != y->col) return x->col < y->col ? -1 : 1;
    if (x->row != y->row) return x->row < y->row ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}

/* contributions of every factor, scalar index of node n = idxs[n] (aprilsam.c:154-195) */
static void accumulate(int F, const int *fa, const int *fb, const double *lp, const double *st_unary,
                       const double *z, const double *W, const int *idxs, trips_t *T, double *B)
{
    double J[2][9], r[3], JtW[9], H[9];
    for (int f = 0; f < F; f++) {
        int nn = fb[f] >= 0 ? 2 : 1;
        int nodes[2] = { fa[f], fb[f] };
        const double *w = W + 9 * f;
        if (nn == 2) orc_factor_eval(1, lp + 3 * fa[f], lp + 3 * fb[f], z + 3 * f, w, J[0], J[1], r);
        else orc_factor_eval(0, st_unary + 3 * fa[f], NULL, z + 3 * f, w, J[0], J[1], r);
        for (int z0 = 0; z0 < nn; z0++) {
            int n0 = nodes[z0];
            mul_atb(J[z0], w, JtW);
            for (int z1 = 0; z1 < nn; z1++) {
                int n1 = nodes[z1];
                mul_ab(JtW, J[z1], H);
                for (int row = 0; row < 3; row++)
                    for (int col = 0; col < 3; col++) {
                        if (row + idxs[n0] > col + idxs[n1]) continue;     /* upper triangle only */
                        trip_add(T, row + idxs[n0], col + idxs[n1], H[3 * row + col]);
                    }
            }
            for (int row = 0; row < 3; row++) {
                double acc = 0;
                for (int k = 0; k < 3; k++) acc += JtW[3 * row + k] * r[k];
                B[idxs[n0] + row] += acc;
            }
        }
    }
}

void orc_normal_equations_dense(int N, const double *lp, const double *st_unary, int F, const int *fa, const int *fb,
                                const double *z, const double *W, double lambda, double *A, double *B)
{
    int n = 3 * N;
    int *idxs = malloc(sizeof(int) * N);
    for (int i = 0; i < N; i++) idxs[i] = 3 * i;
    trips_t T = { 0 };
    memset(B, 0, sizeof(double) * n);
    memset(A, 0, sizeof(double) * n * n);
    accumulate(F, fa, fb, lp, st_unary, z, W, idxs, &T, B);
    for (long k = 0; k < T.n; k++) A[(long)T.t[k].row * n + T.t[k].col] += T.t[k].v;
    if (lambda > 0) for (int i = 0; i < n; i++) A[(long)i * n + i] += lambda;
    free(T.t); free(idxs);
}

/* ---- node-level greedy minimum degree ------------------------------------------------------------- */
typedef struct { int *nb; int n, cap; } nbrs_t;
static void nb_push(nbrs_t *a, int v)
{
    if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 8; a->nb = realloc(a->nb, sizeof(int) * a->cap); }
    a->nb[a->n++] = v;
}
typedef struct { int deg, id; } hent_t;
static void heap_push(hent_t *h, long *n, hent_t e)
{
    long i = (*n)++;
    h[i] = e;
    while (i > 0) {
        long p = (i - 1) / 2;
        if (h[p].deg < h[i].deg || (h[p].deg == h[i].deg && h[p].id < h[i].id)) break;
        hent_t t = h[p]; h[p] = h[i]; h[i] = t; i = p;
    }
}
static hent_t heap_pop(hent_t *h, long *n)
{
    hent_t top = h[0];
    h[0] = h[--(*n)];
    long i = 0;
    for (;;) {
        long l = 2 * i + 1, r = l + 1, m = i;
        if (l < *n && (h[l].deg < h[m].deg || (h[l].deg == h[m].deg && h[l].id < h[m].id))) m = l;
        if (r < *n && (h[r].deg < h[m].deg || (h[r].deg == h[m].deg && h[r].id < h[m].id))) m = r;
        if (m == i) break;
        hent_t t = h[m]; h[m] = h[i]; h[i] = t; i = m;
    }
    return top;
}
static void min_degree_order(int N, int F, const int *fa, const int *fb, int *order)
{
    nbrs_t *g = calloc(N, sizeof(nbrs_t));
    int *mark = calloc(N, sizeof(int)), token = 0;
    char *gone = calloc(N, 1);
    for (int f = 0; f < F; f++) if (fb[f] >= 0 && fa[f] != fb[f]) { nb_push(&g[fa[f]], fb[f]); nb_push(&g[fb[f]], fa[f]); }
    for (int i = 0; i < N; i++) {                    /* dedupe */
        token++; int w = 0;
        for (int k = 0; k < g[i].n; k++) { int v = g[i].nb[k]; if (mark[v] != token) { mark[v] = token; g[i].nb[w++] = v; } }
        g[i].n = w;
    }
    long hcap = 64 + 8L * N, hn = 0;
    hent_t *heap = malloc(sizeof(hent_t) * hcap);
    for (int i = 0; i < N; i++) { hent_t e = { g[i].n, i }; heap_push(heap, &hn, e); }
    int k = 0;
    while (k < N) {
        hent_t e = heap_pop(heap, &hn);
        if (gone[e.id] || e.deg != g[e.id].n) continue;       /* stale entry */
        int b = e.id;
        order[k++] = b; gone[b] = 1;
        /* marginalise b: its neighbours become a clique */
        for (int ai = 0; ai < g[b].n; ai++) {
            int a = g[b].nb[ai];
            nbrs_t *na = &g[a];
            token++;
            int w = 0;
            for (int q = 0; q < na->n; q++) { int v = na->nb[q]; if (v == b) continue; mark[v] = token; na->nb[w++] = v; }
            na->n = w;
            mark[a] = token;
            for (int bi = 0; bi < g[b].n; bi++) { int v = g[b].nb[bi]; if (mark[v] != token) { mark[v] = token; nb_push(na, v); } }
            if (hn + 1 >= hcap) { hcap *= 2; heap = realloc(heap, sizeof(hent_t) * hcap); }
            hent_t ne = { na->n, a }; heap_push(heap, &hn, ne);
        }
    }
    for (int i = 0; i < N; i++) free(g[i].nb);
    free(g); free(mark); free(gone); free(heap);
}

/* ---- up-looking sparse Cholesky on the upper-triangular CSC (Ap, Ai, Ax), n columns ------------ */
static void etree_upper(int n, const int *Ap, const int *Ai, int *parent)
{
    int *anc = malloc(sizeof(int) * n);
    for (int k = 0; k < n; k++) {
        parent[k] = -1; anc[k] = -1;
        for (int p = Ap[k]; p < Ap[k + 1]; p++) {
            int i = Ai[p];
            while (i != -1 && i < k) {
                int next = anc[i];
                anc[i] = k;
                if (next == -1) parent[i] = k;
                i = next;
            }
        }
    }
    free(anc);
}
/* nonzero pattern of row k of L: s[top..n-1], topologically ordered; flag[] is restored via w marks */
static int row_reach(int n, const int *Ap, const int *Ai, int k, const int *parent, int *s, int *w)
{
    int top = n;
    w[k] = k;
    for (int p = Ap[k]; p < Ap[k + 1]; p++) {
        int i = Ai[p];
        if (i > k) continue;
        int len = 0;
        for (; w[i] != k; i = parent[i]) { s[len++] = i; w[i] = k; }
        while (len > 0) s[--top] = s[--len];
    }
    return top;
}

/* order_in (position -> node, N ints) or NULL.  The solution of the normal equations does not depend on the elimination order -- but
 * the normal equations themselves do when a factor's W is not symmetric AS GIVEN: only entries in the upper triangle of the ORDERED
 * matrix are accumulated (aprilsam.c:171), so a factor (a, b) contributes J_a'W J_b when a is eliminated first and J_b'W J_a
 * otherwise, which are transposes of each other only for a symmetric W.  For such inputs the reference's own order (param->ordering,
 * aprilsam.c:999-1249; held in the golden fixtures) has to be supplied. */
int orc_solve_system_ordered(int N, const double *lp, const double *st_unary, int F, const int *fa, const int *fb,
                             const double *z, const double *W, const double *lambda_node, double *dx,
                             const int *order_in, int *ordering_out, double *stats)
{
    const int n = 3 * N;
    int *order = malloc(sizeof(int) * N), *idxs = malloc(sizeof(int) * N);
    if (order_in) memcpy(order, order_in, sizeof(int) * N);
    else min_degree_order(N, F, fa, fb, order);
    for (int i = 0; i < N; i++) idxs[order[i]] = 3 * i;                 /* aprilsam.c:141-148 */
    if (ordering_out) memcpy(ordering_out, order, sizeof(int) * N);

    trips_t T = { 0 };
    double *B = calloc(n, sizeof(double));
    accumulate(F, fa, fb, lp, st_unary, z, W, idxs, &T, B);
    for (int i = 0; i < N; i++)                                          /* Tikhonov, aprilsam.c:197-204 */
        for (int d = 0; d < 3; d++) trip_add(&T, idxs[i] + d, idxs[i] + d, lambda_node ? lambda_node[i] : 0.0);
    qsort(T.t, T.n, sizeof(trip_t), trip_cmp);
    /* compress to CSC (upper), summing duplicates in insertion order */
    int *Ap = calloc(n + 1, sizeof(int)), *Ai = malloc(sizeof(int) * (T.n + 1));
    double *Ax = malloc(sizeof(double) * (T.n + 1));
    long nz = 0;
    for (long k = 0; k < T.n; k++) {
        if (k > 0 && T.t[k].col == T.t[k - 1].col && T.t[k].row == T.t[k - 1].row) { Ax[nz - 1] += T.t[k].v; continue; }
        Ai[nz] = T.t[k].row; Ax[nz] = T.t[k].v; Ap[T.t[k].col + 1]++; nz++;
    }
    for (int j = 0; j < n; j++) Ap[j + 1] += Ap[j];
    free(T.t);

    int *parent = malloc(sizeof(int) * n), *s = malloc(sizeof(int) * n), *w = malloc(sizeof(int) * n);
    etree_upper(n, Ap, Ai, parent);
    /* symbolic: column counts of L through the row patterns */
    int *cnt = calloc(n + 1, sizeof(int));
    for (int i = 0; i < n; i++) w[i] = -1;
    for (int k = 0; k < n; k++) {
        int top = row_reach(n, Ap, Ai, k, parent, s, w);
        for (int q = top; q < n; q++) cnt[s[q]]++;
        cnt[k]++;
    }
    long *Lp = malloc(sizeof(long) * (n + 1));
    Lp[0] = 0;
    double sumsq = 0;
    for (int j = 0; j < n; j++) { Lp[j + 1] = Lp[j] + cnt[j]; sumsq += (double)cnt[j] * cnt[j]; }
    if (stats) { stats[0] = (double)Lp[n]; stats[1] = sumsq; }
    int *Li = malloc(sizeof(int) * Lp[n]);
    double *Lx = malloc(sizeof(double) * Lp[n]);
    long *fill = malloc(sizeof(long) * n);
    double *x = calloc(n, sizeof(double));
    for (int j = 0; j < n; j++) fill[j] = Lp[j];
    for (int i = 0; i < n; i++) w[i] = -1;
    int rc = 0;
    for (int k = 0; k < n && rc == 0; k++) {
        int top = row_reach(n, Ap, Ai, k, parent, s, w);
        x[k] = 0;
        for (int p = Ap[k]; p < Ap[k + 1]; p++) if (Ai[p] <= k) x[Ai[p]] = Ax[p];
        double d = x[k];
        x[k] = 0;
        for (int q = top; q < n; q++) {
            int i = s[q];
            double lki = x[i] / Lx[Lp[i]];
            x[i] = 0;
            for (long p = Lp[i] + 1; p < fill[i]; p++) x[Li[p]] -= Lx[p] * lki;
            d -= lki * lki;
            long p = fill[i]++;
            Li[p] = k; Lx[p] = lki;
        }
        if (d <= 0) { rc = -1; break; }                                  /* csparse.c:505-506 */
        long p = fill[k]++;
        Li[p] = k; Lx[p] = sqrt(d);
    }
    if (rc == 0) {
        /* L y = B (== U'y = B), then L' x = y */
        double *y = B;
        for (int j = 0; j < n; j++) {
            y[j] /= Lx[Lp[j]];
            for (long p = Lp[j] + 1; p < Lp[j + 1]; p++) y[Li[p]] -= Lx[p] * y[j];
        }
        for (int j = n - 1; j >= 0; j--) {
            for (long p = Lp[j] + 1; p < Lp[j + 1]; p++) y[j] -= Lx[p] * y[Li[p]];
            y[j] /= Lx[Lp[j]];
        }
        for (int i = 0; i < N; i++) for (int d = 0; d < 3; d++) dx[3 * i + d] = y[idxs[i] + d];
    }
    free(order); free(idxs); free(B); free(Ap); free(Ai); free(Ax); free(parent); free(s); free(w); free(cnt);
    free(Lp); free(Li); free(Lx); free(fill); free(x);
    return rc;
}

int orc_solve_system(int N, const double *lp, const double *st_unary, int F, const int *fa, const int *fb,
                     const double *z, const double *W, const double *lambda_node, double *dx,
                     int *ordering_out, double *stats)
{
    return orc_solve_system_ordered(N, lp, st_unary, F, fa, fb, z, W, lambda_node, dx, NULL, ordering_out, stats);
}

int orc_batch_step(int N, double *states, int F, const int *fa, const int *fb, const double *z, const double *W,
                   double lambda, double *dx_out, double *stats)
{
    return orc_batch_step_ordered(N, states, F, fa, fb, z, W, lambda, NULL, dx_out, stats);
}

int orc_batch_step_ordered(int N, double *states, int F, const int *fa, const int *fb, const double *z, const double *W,
                           double lambda, const int *order_in, double *dx_out, double *stats)
{
    double *lp = malloc(sizeof(double) * 3 * N), *dx = malloc(sizeof(double) * 3 * N), *lam = malloc(sizeof(double) * N);
    memcpy(lp, states, sizeof(double) * 3 * N);                          /* relinearize, aprilsam.c:131-135 */
    for (int i = 0; i < N; i++) lam[i] = lambda > 0 ? lambda : 0;
    int rc = orc_solve_system_ordered(N, lp, lp, F, fa, fb, z, W, lam, dx, order_in, NULL, stats);
    if (rc == 0) {
        for (int i = 0; i < N; i++) {                                    /* april_graph_xyt.c:302-314 */
            const double *d = dx + 3 * i;
            if (isnan(d[0]) || isnan(d[1]) || isnan(d[2])) continue;
            states[3 * i] = lp[3 * i] + d[0];
            states[3 * i + 1] = lp[3 * i + 1] + d[1];
            states[3 * i + 2] = orc_mod2pi(lp[3 * i + 2] + d[2]);
        }
        if (dx_out) memcpy(dx_out, dx, sizeof(double) * 3 * N);
    }
    free(lp); free(dx); free(lam);
    return rc;
}

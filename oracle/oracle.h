/* oracle.h — CPU restatement of the reference algorithm for the april_graph_cholesky path.
 * TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it. */
#ifndef APRILSAM_ORACLE_H
#define APRILSAM_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

double orc_mod2pi(double v);

/* one factor: binary != 0 -> xyt (pa, pb), else xytpos (pa). Row-major 3x3 J0, J1; r[3]; returns r'Wr */
double orc_factor_eval(int binary, const double *pa, const double *pb, const double *z, const double *W,
                       double *J0, double *J1, double *r);

/* chi^2 at `states` with the 1/2-on-xyt-only convention. fb[i] < 0 marks an xytpos factor. */
double orc_chi2(int N, const double *states, int F, const int *fa, const int *fb, const double *z, const double *W);

/* Solve (sum J'WJ + diag(lambda_node)) dx = sum J'W r with xyt factors linearised at lp and xytpos
 * factors at st_unary.  dx: 3 per node, node order.  ordering_out (N ints, may be NULL) receives the
 * elimination order used.  stats (may be NULL): [0] nnz(L), [1] sum_j c_j^2.  Returns 0, or -1 if a
 * non-positive pivot was met. */
int orc_solve_system(int N, const double *lp, const double *st_unary, int F, const int *fa, const int *fb,
                     const double *z, const double *W, const double *lambda_node, double *dx,
                     int *ordering_out, double *stats);

/* The same with the elimination order given (position -> node; NULL = the oracle's own min-degree order).  Needed only when some W is
 * not symmetric as given: the reference accumulates the upper triangle of the ORDERED matrix with W as given (aprilsam.c:171), so its
 * normal equations then depend on its order (param->ordering, held in the golden fixtures). */
int orc_solve_system_ordered(int N, const double *lp, const double *st_unary, int F, const int *fa, const int *fb,
                             const double *z, const double *W, const double *lambda_node, double *dx,
                             const int *order_in, int *ordering_out, double *stats);
int orc_batch_step_ordered(int N, double *states, int F, const int *fa, const int *fb, const double *z, const double *W,
                           double lambda, const int *order_in, double *dx_out, double *stats);

/* One batch Gauss-Newton step: l_point <- states; solve; states <- l_point + dx (theta wrapped, NaN rows
 * skipped).  Returns 0 / -1 as above. */
int orc_batch_step(int N, double *states, int F, const int *fa, const int *fb, const double *z, const double *W,
                   double lambda, double *dx_out, double *stats);

/* Upper-triangle normal equations in NODE/DoF coordinates (scalar index 3*node + dof), as the reference
 * accumulates them (aprilsam.c:159-204) but un-permuted: dense row-major n x n (n = 3N) for small N. */
void orc_normal_equations_dense(int N, const double *lp, const double *st_unary, int F, const int *fa, const int *fb,
                                const double *z, const double *W, double lambda, double *A, double *B);

#ifdef __cplusplus
}
#endif
#endif

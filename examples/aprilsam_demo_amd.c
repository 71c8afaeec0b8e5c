/* aprilsam_demo_amd.c — counterpart of the reference's examples/aprilsam_demo.c for libaprilsam_amd.so.
 *
 * Same simulation (examples/aprilsam_demo.c:119-234): poses of a VERTEX2/EDGE2 text file (data/M3500.txt format,
 * :52-99) are added one by one; pose 0 gets the prior W = diag(1e4, 1e4, 1e3) and a batch step, every later pose
 * copies the loaded factors whose max node id equals the pose id, seeds the new pose from "odom" factors
 * (|a-b| == 1) and runs april_graph_cholesky_inc (or april_graph_cholesky with --batch_update_only); chi^2 and
 * the solver time are printed per step exactly like the reference does.  Same flags:
 *     --datapath FILE  --batch_update_only  --nthreshold N  --delta_xy X  --delta_theta T   [--quiet] [--max_poses N]
 *
 * Plain C against include/aprilsam_amd.h only:
 *     gcc -O2 -Iinclude examples/aprilsam_demo_amd.c -Laprilsam_amd/lib -laprilsam_amd -Wl,-rpath,$PWD/aprilsam_amd/lib -lm -o aprilsam_demo_amd
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "aprilsam_amd.h"

typedef struct { int a, b; double z[3], W[9]; int odom; } edge_t;

static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static void xyt_mul(const double *a, const double *b, double *r)   /* common/doubles_floats_impl.h:498-506 */
{ double s = sin(a[2]), c = cos(a[2]); r[0] = c * b[0] - s * b[1] + a[0]; r[1] = s * b[0] + c * b[1] + a[1]; r[2] = a[2] + b[2]; }
static void xyt_inv(const double *a, double *r)                     /* :569-575 */
{ double s = sin(a[2]), c = cos(a[2]); r[0] = -s * a[1] - c * a[0]; r[1] = -c * a[1] + s * a[0]; r[2] = -a[2]; }

int main(int argc, char **argv)
{
    const char *path = NULL; int batch_only = 0, nthreshold = 100, quiet = 0, max_poses = -1;
    double delta_xy = 0.1, delta_theta = 0.1;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--datapath") && i + 1 < argc) path = argv[++i];
        else if (!strcmp(argv[i], "--batch_update_only")) batch_only = 1;
        else if (!strcmp(argv[i], "--nthreshold") && i + 1 < argc) nthreshold = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--delta_xy") && i + 1 < argc) delta_xy = atof(argv[++i]);
        else if (!strcmp(argv[i], "--delta_theta") && i + 1 < argc) delta_theta = atof(argv[++i]);
        else if (!strcmp(argv[i], "--max_poses") && i + 1 < argc) max_poses = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--quiet")) quiet = 1;
        else { fprintf(stderr, "usage: %s --datapath FILE [--batch_update_only] [--nthreshold N] [--delta_xy X] [--delta_theta T] [--max_poses N] [--quiet]\n", argv[0]); return 1; }
    }
    if (!path) { fprintf(stderr, "--datapath is required (VERTEX2 / EDGE2 text file)\n"); return 1; }
    FILE *f = fopen(path, "r");
    if (!f) { perror(path); return 1; }
    int nv = 0, ne = 0, capv = 1024, cape = 2048;
    double (*init)[3] = malloc(sizeof(double[3]) * capv);
    edge_t *edges = malloc(sizeof(edge_t) * cape);
    char tok[32];
    while (fscanf(f, "%31s", tok) == 1) {
        if (!strcmp(tok, "VERTEX2")) {
            int id; if (nv == capv) init = realloc(init, sizeof(double[3]) * (capv *= 2));
            if (fscanf(f, "%d %lf %lf %lf", &id, &init[nv][0], &init[nv][1], &init[nv][2]) != 4) return 2;
            nv++;
        } else if (!strcmp(tok, "EDGE2")) {          /* IDout IDin dx dy dth I11 I12 I22 I33 I13 I23 */
            if (ne == cape) edges = realloc(edges, sizeof(edge_t) * (cape *= 2));
            edge_t *e = &edges[ne]; double *W = e->W; memset(W, 0, sizeof(e->W));
            if (fscanf(f, "%d %d %lf %lf %lf %lf %lf %lf %lf %lf %lf", &e->a, &e->b, &e->z[0], &e->z[1], &e->z[2],
                       &W[0], &W[1], &W[4], &W[8], &W[2], &W[5]) != 11) return 2;
            e->odom = abs(e->b - e->a) == 1;
            ne++;
        } else { fprintf(stderr, "unexpected token %s\n", tok); return 2; }
    }
    fclose(f);
    printf("%d nodes,  factors: %d \n", nv, ne);
    if (max_poses > 0 && max_poses < nv) nv = max_poses;

    april_graph_cholesky_param_t *param = calloc(1, sizeof(*param));
    april_graph_cholesky_param_init(param);
    param->delta_xy = delta_xy; param->delta_theta = delta_theta; param->nthreshold = nthreshold;
    april_graph_t *graph = april_graph_create();
    matd_t *W = calloc(1, sizeof(matd_t) + 9 * sizeof(double)); W->nrows = W->ncols = 3;
    double total = 0;
    for (int k = 0; k < nv; k++) {
        if (!quiet) printf("Step: %d / %d \n", k, nv);
        april_graph_node_t *node = april_graph_node_xyt_create(init[k], init[k], init[k]);
        aprilsam_amd_graph_add_node(graph, node);
        if (k == 0) {
            double z0[3] = { 0, 0, 0 };
            memset(W->data, 0, 72); W->data[0] = 10000; W->data[4] = 10000; W->data[8] = 1000;
            aprilsam_amd_graph_add_factor(graph, april_graph_factor_xytpos_create(0, z0, NULL, W));
        } else {
            for (int i = 0; i < ne; i++) {
                edge_t *e = &edges[i];
                int mx = e->a > e->b ? e->a : e->b;
                if (mx != k) continue;
                april_graph_node_t **ns = (april_graph_node_t **)graph->nodes->data;
                if (e->odom) {
                    if (e->a < e->b) { xyt_mul(ns[e->a]->state, e->z, ns[e->b]->state); ns[e->b]->relinearize(ns[e->b]); }
                    else { double iz[3]; xyt_inv(e->z, iz); xyt_mul(ns[e->b]->state, iz, ns[e->a]->state); ns[e->a]->relinearize(ns[e->a]); }
                }
                memcpy(W->data, e->W, 72);
                aprilsam_amd_graph_add_factor(graph, april_graph_factor_xyt_create(e->a, e->b, e->z, NULL, W));
            }
        }
        double t0 = now_ms();
        if (batch_only || k == 0) april_graph_cholesky(graph, param);
        else april_graph_cholesky_inc(graph, param);
        double step = now_ms() - t0;
        total += step;
        double chi2 = april_graph_chi2(graph);
        if (!quiet || k == nv - 1)
            printf("Chi squared error: %f \nStep running time: %.3f ms, Total running time: %.3f ms \n", chi2, step, total);
    }
    free(W); free(init); free(edges);
    april_graph_cholesky_param_destory(param);
    april_graph_destroy(graph);
    return 0;
}

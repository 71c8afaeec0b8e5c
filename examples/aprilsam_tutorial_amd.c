/* aprilsam_tutorial_amd.c — counterpart of the reference's examples/aprilsam_tutorial.c for libaprilsam_amd.so.
 *
 * The scenario of examples/aprilsam_tutorial.c:80-266: six poses on a line, one metre apart.  Pose 0 gets the prior
 * W = diag(1e4, 1e4, 1e3) and a batch step; every later pose k arrives with an odometry factor (k-1, k) with
 * sigma = (0.1 m, 0.1 m, 1 deg); the last step adds a loop closure (0, 5) that claims pose 5 sits at (5, 1, 0).
 * After each arrival the graph is optimised (april_graph_cholesky for the first / with --batch_update_only,
 * april_graph_cholesky_inc otherwise) and chi^2, the timings and every node's state are printed in the reference's
 * format (:67-76).  Flags as in the reference (:278-282):
 *     --nthreshold N  --delta_xy X  --delta_theta T  --batch_update_only
 *
 * Plain C against include/aprilsam_amd.h only:
 *     gcc -O2 -Iinclude examples/aprilsam_tutorial_amd.c -Laprilsam_amd/lib -laprilsam_amd -Wl,-rpath,$PWD/aprilsam_amd/lib -lm -o aprilsam_tutorial_amd
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "aprilsam_amd.h"

static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
/* z = a^-1 * b (common/doubles_floats_impl.h:550-567) */
static void xyt_inv_mul(const double *a, const double *b, double *z)
{
    const double c = cos(a[2]), s = sin(a[2]), dx = b[0] - a[0], dy = b[1] - a[1];
    z[0] = c * dx + s * dy; z[1] = -s * dx + c * dy; z[2] = b[2] - a[2];
}

typedef struct { april_graph_t *graph; april_graph_cholesky_param_t *param; int batch_only; double total; int step; } app_t;

static void optimise_and_report(app_t *app, int first)
{
    const double t0 = now_ms();
    if (first || app->batch_only) april_graph_cholesky(app->graph, app->param);
    else april_graph_cholesky_inc(app->graph, app->param);
    const double dt = now_ms() - t0;
    app->total += dt;
    printf("\n==================== Step: %d ======================= \n", app->step++);
    printf("Chi squared error: %f \nStep running time: %.3f ms, Total running time: %.3f ms \n", april_graph_chi2(app->graph), dt, app->total);
    april_graph_node_t **ns = (april_graph_node_t **)app->graph->nodes->data;
    for (int i = 0; i < app->graph->nodes->size; i++)
        printf("node_%d = {%.2f, %.2f, %.2f} \n", ns[i]->UID, ns[i]->state[0], ns[i]->state[1], ns[i]->state[2]);
}

int main(int argc, char **argv)
{
    app_t app; memset(&app, 0, sizeof(app));
    int nthreshold = 100; double delta_xy = 0.1, delta_theta = 0.1;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--batch_update_only")) app.batch_only = 1;
        else if (!strcmp(argv[i], "--nthreshold") && i + 1 < argc) nthreshold = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--delta_xy") && i + 1 < argc) delta_xy = atof(argv[++i]);
        else if (!strcmp(argv[i], "--delta_theta") && i + 1 < argc) delta_theta = atof(argv[++i]);
        else { fprintf(stderr, "usage: %s [--batch_update_only] [--nthreshold N] [--delta_xy X] [--delta_theta T]\n", argv[0]); return 1; }
    }
    app.param = calloc(1, sizeof(*app.param));
    april_graph_cholesky_param_init(app.param);
    app.param->nthreshold = nthreshold; app.param->delta_xy = delta_xy; app.param->delta_theta = delta_theta;
    app.graph = april_graph_create();

    matd_t *W = calloc(1, sizeof(matd_t) + 9 * sizeof(double)); W->nrows = W->ncols = 3;
    const double deg = 3.14159265358979323846 / 180.0;
    double pose[3] = { 0, 0, 0 }, z0[3] = { 0, 0, 0 };
    aprilsam_amd_graph_add_node(app.graph, april_graph_node_xyt_create(pose, pose, pose));
    W->data[0] = 10000; W->data[4] = 10000; W->data[8] = 1000;
    aprilsam_amd_graph_add_factor(app.graph, april_graph_factor_xytpos_create(0, z0, NULL, W));
    optimise_and_report(&app, 1);

    memset(W->data, 0, 72);
    W->data[0] = 1.0 / (0.1 * 0.1); W->data[4] = 1.0 / (0.1 * 0.1); W->data[8] = 1.0 / (deg * deg);
    for (int k = 1; k < 6; k++) {
        double prev[3] = { k - 1, 0, 0 }, cur[3] = { k, 0, 0 }, z[3];
        aprilsam_amd_graph_add_node(app.graph, april_graph_node_xyt_create(cur, cur, cur));
        xyt_inv_mul(prev, cur, z);
        aprilsam_amd_graph_add_factor(app.graph, april_graph_factor_xyt_create(k - 1, k, z, NULL, W));
        if (k == 5) {                                   /* loop closure: "pose 5 is at (5, 1, 0) seen from pose 0" */
            double origin[3] = { 0, 0, 0 }, claimed[3] = { 5, 1, 0 };
            xyt_inv_mul(origin, claimed, z);
            aprilsam_amd_graph_add_factor(app.graph, april_graph_factor_xyt_create(0, 5, z, NULL, W));
        }
        optimise_and_report(&app, 0);
    }
    free(W);
    april_graph_cholesky_param_destory(app.param);
    april_graph_destroy(app.graph);
    return 0;
}

/* aprilsam_demo_amd.c — counterpart of the reference's examples/aprilsam_demo.c for libaprilsam_amd.so.
 *
 * Same simulation (examples/aprilsam_demo.c:119-234): poses of a VERTEX2/EDGE2 text file (data/M3500.txt format,
 * :52-99) are added one by one; pose 0 gets the prior W = diag(1e4, 1e4, 1e3) and a batch step, every later pose
 * copies the loaded factors whose max node id equals the pose id, seeds the new pose from "odom" factors
 * (|a-b| == 1) and runs april_graph_cholesky_inc (or april_graph_cholesky with --batch_update_only); chi^2 and
 * the solver time are printed per step exactly like the reference does.  The input is either the text file
 * (--datapath, tagged "odom"/"scan" like :83-87) or a `.graph` file (--graphpath, the reference's default input,
 * :249,262) read with april_graph_create_from_file.  Same flags:
 *     --datapath FILE | --graphpath FILE   --batch_update_only  --nthreshold N  --delta_xy X  --delta_theta T
 *     [--savepath FILE: write the loaded graph as a .graph file, what the reference does to /tmp/loaded.graph, :274]
 *     [--quiet] [--max_poses N]
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

static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static void xyt_mul(const double *a, const double *b, double *r)   /* common/doubles_floats_impl.h:498-506 */
{ double s = sin(a[2]), c = cos(a[2]); r[0] = c * b[0] - s * b[1] + a[0]; r[1] = s * b[0] + c * b[1] + a[1]; r[2] = a[2] + b[2]; }
static void xyt_inv(const double *a, double *r)                     /* :569-575 */
{ double s = sin(a[2]), c = cos(a[2]); r[0] = -s * a[1] - c * a[0]; r[1] = -c * a[1] + s * a[0]; r[2] = -a[2]; }

/* examples/aprilsam_demo.c:52-99: VERTEX2 id x y t -> node (state = init = truth); EDGE2 a b dx dy dt I11 I12 I22 I33 I13 I23 ->
 * xyt factor, upper half of W only, tagged "odom" when |a - b| == 1, else "scan" */
static april_graph_t *load_text(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) { perror(path); return NULL; }
    april_graph_t *g = april_graph_create();
    matd_t *W = calloc(1, sizeof(matd_t) + 9 * sizeof(double)); W->nrows = W->ncols = 3;
    char tok[32];
    while (fscanf(f, "%31s", tok) == 1) {
        if (!strcmp(tok, "VERTEX2")) {
            int id; double p[3];
            if (fscanf(f, "%d %lf %lf %lf", &id, &p[0], &p[1], &p[2]) != 4) goto bad;
            aprilsam_amd_graph_add_node(g, april_graph_node_xyt_create(p, p, p));
        } else if (!strcmp(tok, "EDGE2")) {
            int a, b; double z[3], *w = W->data; memset(w, 0, 72);
            if (fscanf(f, "%d %d %lf %lf %lf %lf %lf %lf %lf %lf %lf", &a, &b, &z[0], &z[1], &z[2], &w[0], &w[1], &w[4], &w[8], &w[2], &w[5]) != 11) goto bad;
            april_graph_factor_t *fac = april_graph_factor_xyt_create(a, b, z, NULL, W);
            aprilsam_amd_attr_put_string(&fac->attr, "type", abs(b - a) == 1 ? "odom" : "scan");
            aprilsam_amd_graph_add_factor(g, fac);
        } else { fprintf(stderr, "unexpected token %s\n", tok); goto bad; }
    }
    fclose(f); free(W);
    return g;
bad:
    fclose(f); free(W); april_graph_destroy(g);
    return NULL;
}

int main(int argc, char **argv)
{
    const char *datapath = "", *graphpath = "../data/M3500.graph", *savepath = NULL;
    int batch_only = 0, nthreshold = 100, quiet = 0, max_poses = -1;
    double delta_xy = 0.1, delta_theta = 0.1;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--datapath") && i + 1 < argc) datapath = argv[++i];
        else if (!strcmp(argv[i], "--graphpath") && i + 1 < argc) graphpath = argv[++i];
        else if (!strcmp(argv[i], "--savepath") && i + 1 < argc) savepath = argv[++i];
        else if (!strcmp(argv[i], "--batch_update_only")) batch_only = 1;
        else if (!strcmp(argv[i], "--nthreshold") && i + 1 < argc) nthreshold = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--delta_xy") && i + 1 < argc) delta_xy = atof(argv[++i]);
        else if (!strcmp(argv[i], "--delta_theta") && i + 1 < argc) delta_theta = atof(argv[++i]);
        else if (!strcmp(argv[i], "--max_poses") && i + 1 < argc) max_poses = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--quiet")) quiet = 1;
        else {
            fprintf(stderr, "usage: %s [--datapath TEXTFILE | --graphpath GRAPHFILE] [--savepath GRAPHFILE] [--batch_update_only] [--nthreshold N] "
                            "[--delta_xy X] [--delta_theta T] [--max_poses N] [--quiet]\n", argv[0]);
            return 1;
        }
    }
    april_graph_t *loaded = strlen(datapath) ? load_text(datapath) : april_graph_create_from_file(graphpath);
    if (!loaded) { fprintf(stderr, "cannot load %s\n", strlen(datapath) ? datapath : graphpath); return 2; }
    if (savepath && !april_graph_save(loaded, savepath)) { fprintf(stderr, "cannot write %s\n", savepath); return 2; }
    int nv = loaded->nodes->size;
    const int ne = loaded->factors->size;
    april_graph_node_t **lnodes = (april_graph_node_t **)loaded->nodes->data;
    april_graph_factor_t **lfactors = (april_graph_factor_t **)loaded->factors->data;
    printf("%d nodes,  factors: %d \n", nv, ne);
    if (max_poses >= 0 && max_poses < nv) nv = max_poses;

    april_graph_cholesky_param_t *param = calloc(1, sizeof(*param));
    april_graph_cholesky_param_init(param);
    param->delta_xy = delta_xy; param->delta_theta = delta_theta; param->nthreshold = nthreshold;
    april_graph_t *graph = april_graph_create();
    matd_t *W = calloc(1, sizeof(matd_t) + 9 * sizeof(double)); W->nrows = W->ncols = 3;
    double total = 0;
    for (int k = 0; k < nv; k++) {
        if (!quiet) printf("Step: %d / %d \n", k, nv);
        aprilsam_amd_graph_add_node(graph, lnodes[k]->copy(lnodes[k]));
        if (k == 0) {
            double z0[3] = { 0, 0, 0 };
            memset(W->data, 0, 72); W->data[0] = 10000; W->data[4] = 10000; W->data[8] = 1000;
            aprilsam_amd_graph_add_factor(graph, april_graph_factor_xytpos_create(0, z0, NULL, W));
        } else {
            for (int i = 0; i < ne; i++) {
                april_graph_factor_t *lf = lfactors[i];
                if (lf->type != APRIL_GRAPH_FACTOR_XYT_TYPE) continue;
                const int a = lf->nodes[0], b = lf->nodes[1], mx = a > b ? a : b;
                if (mx != k) continue;
                april_graph_factor_t *fac = lf->copy(lf);                      /* the copy keeps "type" */
                const char *type = aprilsam_amd_attr_get_string(fac->attr, "type");
                april_graph_node_t **ns = (april_graph_node_t **)graph->nodes->data;
                if (type && !strcmp(type, "odom")) {
                    const double *z = fac->u.common.z;
                    if (a < b) { xyt_mul(ns[a]->state, z, ns[b]->state); ns[b]->relinearize(ns[b]); }
                    else { double iz[3]; xyt_inv(z, iz); xyt_mul(ns[b]->state, iz, ns[a]->state); ns[a]->relinearize(ns[a]); }
                }
                aprilsam_amd_graph_add_factor(graph, fac);
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
    free(W);
    april_graph_cholesky_param_destory(param);
    april_graph_destroy(graph);
    april_graph_destroy(loaded);
    return 0;
}

/* aprilsam_amd.h — C-ABI of libaprilsam_amd.so: the MI355X-native replacement for AprilSAM's
 * Gauss-Newton hot path (april_graph_cholesky / april_graph_cholesky_inc).
 *
 * The reference host program keeps compiling against its own aprilsam/aprilsam.h and simply links
 * (or dlopens) this library instead of libaprilsam.so for the symbols declared in PART 2.  PART 1
 * restates — from the measured LP64 layout, SURVEY.md §8(b) — the structs that cross the boundary,
 * so that this library, its tests and its Python mirror agree byte-for-byte with objects created by
 * reference-compiled code.  Nothing here carries torch or HIP types: plain pointers and sizes only.
 *
 * Citations are relative to the reference tree (xipengwang/AprilSAM @ v1).
 */
#ifndef APRILSAM_AMD_H
#define APRILSAM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * PART 1 — boundary structs (layout-compatible restatements)
 * ---------------------------------------------------------------------------------------------- */

/* common/zarray.h:44-51 — 24 bytes; graph->nodes / graph->factors hold POINTERS as elements. */
typedef struct zarray {
    size_t el_sz;
    int    size;
    int    alloc;
    char  *data;
} zarray_t;

/* common/matd.h:46-51 — row-major dense matrix with a flexible data tail. */
typedef struct matd {
    unsigned int nrows, ncols;
    double       data[];
} matd_t;

/* aprilsam.h:65-72 — 32 bytes. attr/stype are opaque to the hot path. */
typedef struct april_graph {
    zarray_t   *factors;   /* elements: april_graph_factor_t*  */
    zarray_t   *nodes;     /* elements: april_graph_node_t*    */
    void       *attr;
    const void *stype;
} april_graph_t;

/* aprilsam.h:75-89 — result of a factor->eval() call (only produced by the host-side vtable
 * entries this library installs on the objects IT creates; the device path never builds one). */
typedef struct april_graph_factor_eval {
    double   chi2;
    matd_t **jacobians;   /* NULL-terminated, one per connected node */
    int      length;
    double  *r;
    matd_t  *W;
} april_graph_factor_eval_t;

#define APRIL_GRAPH_FACTOR_XYT_TYPE    1     /* aprilsam.h:91 */
#define APRIL_GRAPH_FACTOR_XYTPOS_TYPE 2     /* aprilsam.h:92 */
#define APRIL_GRAPH_NODE_XYT_TYPE      100   /* aprilsam.h:94 */

typedef struct april_graph_factor april_graph_factor_t;
typedef struct april_graph_node   april_graph_node_t;

/* aprilsam.h:98-146 — 104 bytes. The device path recognises type 1 (xyt) and 2 (xytpos) and reads
 * nodes[], u.common.z and u.common.W directly; function pointers are never called on the device.
 * Any other type (1 or 2 nodes) is evaluated on the HOST through ->eval(), as aprilsam.c:156 does. */
struct april_graph_factor {
    int   type;
    int   nnodes;
    int  *nodes;
    int   length;
    void *attr;
    april_graph_factor_t      *(*copy)(april_graph_factor_t *factor);
    april_graph_factor_eval_t *(*eval)(april_graph_factor_t *factor, april_graph_t *graph,
                                       april_graph_factor_eval_t *eval);
    april_graph_factor_eval_t *(*state_eval)(april_graph_factor_t *factor, april_graph_t *graph,
                                             april_graph_factor_eval_t *eval);
    void (*destroy)(april_graph_factor_t *factor);
    union {
        struct { double *z; double *ztruth; matd_t *W; void *impl; } common;
        struct { april_graph_factor_t **factors; double *logw; int nfactors; } max;
        struct { void *impl; } impl;
    } u;
    const void *stype;
};

/* aprilsam.h:151-179 — 112 bytes. */
struct april_graph_node {
    int     UID;
    int     type;
    int     length;
    double *state;
    double *init;
    double *truth;
    double *l_point;
    double *delta_X;
    void   *attr;
    april_graph_node_t *(*copy)(april_graph_node_t *node);
    void (*update)(april_graph_node_t *node, double *dstate);
    void (*relinearize)(april_graph_node_t *node);
    void (*destroy)(april_graph_node_t *node);
    void       *impl;
    const void *stype;
};

/* aprilsam.h:231-265 — 128 bytes.  chol / A / tr are the reference's CPU solver state; this library
 * keeps its (device) solver state in a side context keyed by the param pointer and leaves those
 * three NULL, so a reference-compiled april_graph_cholesky_param_destory() stays safe. `ordering`
 * is always a malloc() block (or NULL); factor_num / nreordering / batch_time keep their meaning. */
typedef struct april_graph_cholesky_param {
    double  tikhanov;      /* lambda added to every diagonal in a batch step (default 1e-4) */
    void   *chol;          /* unused here (NULL) */
    int     factor_num;    /* #factors folded into the current factorisation */
    int    *ordering;      /* position -> node id of the current elimination order */
    int     nreordering;   /* #nodes in the current factorisation; must be non-zero on entry */
    int     show_timing;
    double *delta_x;       /* kept only if pre-allocated by the caller (aprilsam.c:363-366) */
    double *B;             /* unused here (NULL) */
    double *y;             /* unused here (NULL) */
    void   *A;             /* unused here (NULL) */
    void   *tr;            /* unused here (NULL) */
    double  l_thresh;
    double  delta_thresh;
    int     nthreshold;
    double  batch_time;
    double  delta_xy;
    double  delta_theta;
} april_graph_cholesky_param_t;

/* ------------------------------------------------------------------------------------------------
 * PART 2 — the drop-in entry points (same names, arguments and error behaviour as the reference)
 * ---------------------------------------------------------------------------------------------- */

/* replaces aprilsam.c:33-39 (aprilsam.h:44): the banner the reference's example programs print first */
void APRILSAM_VERSION(void);
/* replaces aprilsam.c:45-64 */
void april_graph_cholesky_param_init(april_graph_cholesky_param_t *param);
/* replaces aprilsam.c:66-85 (frees the side context, the owned arrays AND param itself) */
void april_graph_cholesky_param_destory(april_graph_cholesky_param_t *param);
/* replaces aprilsam.c:87-375 — one batch Gauss-Newton step on the GPU; synchronous: every
 * node->state / l_point / delta_X is valid in host memory on return. Silent no-op on an empty graph. */
void april_graph_cholesky(april_graph_t *graph, april_graph_cholesky_param_t *param);
/* replaces aprilsam.c:377-576 — incremental step (new nodes/factors since the last call). */
void april_graph_cholesky_inc(april_graph_t *graph, april_graph_cholesky_param_t *param);
/* replaces aprilsam.c:578-597 — solve + state update of the current incremental factorisation. */
void april_graph_cholesky_inc_solver(april_graph_t *graph, april_graph_cholesky_param_t *param, int *idxs);
/* replaces april_graph.c:79-98 — chi^2 with the reference's 1/2-on-xyt-only convention (GPU). */
double april_graph_chi2(april_graph_t *graph);

/* ------------------------------------------------------------------------------------------------
 * PART 3 — host-side object constructors with the reference's names and ABI, so a caller (or a
 * test) can build a graph without the reference library.  replaces april_graph.c:329-364,
 * april_graph_xyt.c:276-298,420-438, april_graph_xytpos.c:191-211, april_graph.c:33-49.
 * ---------------------------------------------------------------------------------------------- */
april_graph_t *april_graph_create(void);
void           april_graph_destroy(april_graph_t *graph);
april_graph_node_t   *april_graph_node_xyt_create(const double *state, const double *init, const double *truth);
april_graph_factor_t *april_graph_factor_xyt_create(int a, int b, const double *z, const double *ztruth, const matd_t *W);
april_graph_factor_t *april_graph_factor_xytpos_create(int a, double *z, double *ztruth, matd_t *W);
void april_graph_factor_eval_destroy(april_graph_factor_eval_t *eval);
int  april_graph_dof(april_graph_t *graph);
/* `.graph` files: replaces april_graph_save / april_graph_create_from_file
 * (april_graph.c:377-426) and the object stream under them (common/stype.c:75-169, encode_bytes.h:120-250) for xyt
 * nodes and xyt / xytpos factors.  save returns 1 on success, 0 on failure (the reference's convention); load
 * returns NULL on failure.  String attributes survive a round trip (see aprilsam_amd_attr_*); attribute values of
 * other types are skipped on input. */
int            april_graph_save(april_graph_t *graph, const char *path);
april_graph_t *april_graph_create_from_file(const char *path);
void           april_graph_stype_init(void);      /* april_graph.c:367-375; nothing to register here */

/* ------------------------------------------------------------------------------------------------
 * PART 4 — extensions (prefix aprilsam_amd_). Not part of the reference API.
 * ---------------------------------------------------------------------------------------------- */

/* zarray_add is `static inline` in the reference (common/zarray.h:180-189), so it is not an
 * exported symbol there either; these two do the same append for callers without that header. */
void aprilsam_amd_graph_add_node(april_graph_t *graph, april_graph_node_t *node);
void aprilsam_amd_graph_add_factor(april_graph_t *graph, april_graph_factor_t *factor);

/* String attributes (the only kind the `.graph` files of the path carry: "type" = "odom" | "scan" on the demo's
 * factors, examples/aprilsam_demo.c:84-86; reference API: april_graph_*_attr_put/get, april_graph.c:101-176).
 * attr_slot is &node->attr, &factor->attr or &graph->attr of an object created by THIS library; a slot already
 * holding the reference library's attribute table is refused (-2).  Objects' copy()/destroy() carry them along. */
int         aprilsam_amd_attr_put_string(void **attr_slot, const char *key, const char *value);
const char *aprilsam_amd_attr_get_string(const void *attr, const char *key);
int         aprilsam_amd_attr_count(const void *attr);
int         aprilsam_amd_attr_item(const void *attr, int i, const char **key, const char **value);

/* Same as april_graph_save / april_graph_create_from_file.  save_ex starts the per-object magic counter of the file
 * format `magic_offset` objects later, which reproduces byte for byte what a reference process writes after it has
 * already encoded that many objects (csrc/graph_io.cpp; data/M3500.graph: 8 * 5453). */
int            aprilsam_amd_graph_save(april_graph_t *graph, const char *path);
int            aprilsam_amd_graph_save_ex(april_graph_t *graph, const char *path, unsigned long long magic_offset);
april_graph_t *aprilsam_amd_graph_load(const char *path);

/* Host-only consistency checks of the index arithmetic shared by launch tables and kernels (tile decode of the
 * outer-blocked trailing update, packed Schur offsets, panel row tiles, LDS budgets).  0 = all good.  No GPU needed. */
int aprilsam_amd_selftest(void);

/* Number of usable HIP devices (0 => every solver entry point fails loudly: error -14, see below). */
int aprilsam_amd_device_count(void);
/* Select the HIP device used by contexts created afterwards (default: LOCAL_RANK env or 0). */
int aprilsam_amd_set_device(int device);
/* Bind ONE param (and the graph it is called with) to a device slot, 0 <= slot < 64: slot s runs on HIP device s % device_count, so on a
 * node with N devices the slots 0 .. N-1 are the devices and further slots share them.  Calls on params bound to different slots run
 * concurrently from different threads (each slot has its own lock, every call makes its slot's device current on the calling thread):
 * one C process can drive N solves on N devices without forking -- the reference's solver has no process-wide state either (SURVEY
 * section 8(b) "Threading").  Calls on the same slot are serialised.  Whatever the param held (plan, fronts, captured graphs) is dropped;
 * a graph's device copies follow the slot of the param it is called with (one graph is driven from one slot at a time).  Options
 * (aprilsam_amd_set_option) stay process-global: set them while no call is in flight.  Returns 0, -1 for a bad slot.
 * The binding lasts until april_graph_cholesky_param_init or _destory of that param (bind AFTER _init, as the example in
 * INTEGRATION.md section 4b does): a param allocated later at the same address starts on the default slot.
 * aprilsam_amd_param_get_device: the HIP device a call on this param runs on. */
int aprilsam_amd_param_set_device(const april_graph_cholesky_param_t *param, int slot);
int aprilsam_amd_param_get_device(const april_graph_cholesky_param_t *param);

/* Per-param solver statistics of the LAST solver call (all times in milliseconds).  */
typedef struct aprilsam_amd_stats {
    int    n_nodes, n_factors;
    int    n_fronts, n_levels;         /* supernodal assembly tree */
    int    max_front_rows;             /* largest frontal dimension (scalar rows, w/o rhs row) */
    int    symbolic_reused;            /* 1 if ordering+symbolic came from the cache */
    int    not_spd;                    /* 1 if a non-positive pivot was met (states left untouched) */
    int    reserved0;
    long long nnz_L;                   /* scalar non-zeros of L incl. diagonal (dense-front count) */
    double flops_factor;               /* sum_j c_j^2 of our own L (SURVEY §8(d) convention) */
    double bytes_fronts;               /* bytes of all frontal matrices resident in HBM */
    double ms_pack, ms_symbolic, ms_h2d, ms_device, ms_d2h, ms_unpack, ms_total;
    double ms_dev_linearize, ms_dev_factor, ms_dev_solve;   /* HIP-event timings inside ms_device */
    double chi2_before;                /* chi^2 at the linearisation point (from the linearise kernel) */
    int    error_code;                 /* 0, or the code of the failure that ended the last call on this param (see below) */
    int    reserved1;          /* 1: a warm batch call launched on the packed factor copies, found an edited factor object afterwards and ran again (speculate_factors) */
    /* last april_graph_cholesky_inc on this param (both 0 after any other call): */
    int    inc_replanned;      /* 1: the step did not fit the frozen plan of the last batch step and was solved on a fresh plan (ordering +
                                  symbolic analysis + every front factorised); the result is the exact solve of the incremental system */
    int    inc_old_old_cross;  /* number of the step's new factors that connect two poses which BOTH predate the call and lie in different
                                  branches of the reference's elimination tree.  For those the reference's partial re-factorisation
                                  (aprilsam.c:850-906, children first over the OLD tree) finalises one row before the other has updated it and
                                  returns something that is NOT the solution of its own normal equations; this library returns the exact solve
                                  (INTEGRATION.md section 4).  > 0 therefore means: this step's states deviate from the reference's by design */
    int    inc_fronts_updated; /* fronts whose factor took a low-rank update in this step (option "inc_update") instead of being re-assembled and
                                  re-factorised; reserved0 counts all fronts the step regenerated */
    int    reserved2;
} aprilsam_amd_stats_t;
int aprilsam_amd_get_stats(const april_graph_cholesky_param_t *param, aprilsam_amd_stats_t *out);

/* Failure path.  The reference's entry points are void and crash on bad input (assert / NULL dereference, SURVEY.md
 * section 8(b)); this library never takes the caller's process down.  A call that fails returns with the caller's node states
 * untouched, prints one line on stderr, and leaves
 *     -2  a pivot was not positive (stats.not_spd = 1; information matrix not positive definite)
 *     -9  a multi-level launch gave up waiting for a dependency flag (should not happen; reported, not hung)
 *    -10  a HIP runtime call failed                  -11  device / pinned memory exhausted (or option "mem_cap_mb")
 *    -12  unsupported input: node type other than xyt, a foreign factor (own eval()) with more than 11 nodes, a factor of a native type
 *         with the wrong number of nodes, an unsplittable dense region of
 *         more than ~6000 poses, more than 22 million factors, param->nreordering == 0 (the reference asserts)
 *    -13  malformed graph: node index out of range, a factor connecting a node to itself, incomplete eval() result
 *    -14  no HIP device visible: there is NO CPU fallback, every solver call on such a machine fails this way (april_graph_chi2
 *         returns NaN) -- nothing is ever computed on the host
 *    -15  internal inconsistency of the planner
 *    -16  debug option "pool_guard": a kernel wrote into the guard band behind a frontal array
 * in stats.error_code and in aprilsam_amd_last_error (most recent failure of the process; msg may be NULL).  The param's
 * cached plan and factorisation are dropped: the next april_graph_cholesky starts from scratch, april_graph_cholesky_inc
 * returns silently until then (no prior factorisation, aprilsam.c:382-383). */
int  aprilsam_amd_last_error(char *msg, int cap);
void aprilsam_amd_clear_error(void);

/* Runtime options (also settable by env APRILSAM_AMD_<NAME>): returns 0 on success.
 *   "leaf_nodes"        nested-dissection leaf size in pose nodes (default 16)
 *   "deterministic"     1 = disable the wall-clock fallback rule aprilsam.c:557-559; default 0 = the reference's behaviour: an incremental step
 *                       that took longer than param->batch_time / 3 falls back to a batch step, so the schedule of fall-backs (and with it
 *                       the states, within the solver's tolerance) depends on the machine's timing exactly as the reference's does.  Parity
 *                       runs and recorded schedules set 1 (env APRILSAM_AMD_DETERMINISTIC=1)
 *   "use_graph"         1 = replay the numeric phase from a captured hipGraph (default 1)
 *   "device_timing"     1 = record per-stage HIP events (default 0)
 *   "trust_factor_cache" 1 = z/W of already-packed factors are treated as immutable: skips the per-call re-read and
 *                       comparison of every factor object (default 0: reference semantics, edits in place are seen)
 *   "small_lds_kb"      LDS budget (KiB) of the single-workgroup front kernel: fronts whose whole array fits run fully
 *                       in LDS, fronts whose own columns fit run in panel mode, the rest takes the multi-workgroup
 *                       path (default 156; 0 forces the multi-workgroup path everywhere)
 *   "panel_mode"        0 = no panel mode (fronts that do not fit LDS entirely go to the multi-workgroup path) (1)
 *   "tp_fronts", "tp_lds_kb"  levels with at least tp_fronts fronts (default 1000) are throughput-bound: there only
 *                       fronts up to tp_lds_kb KiB (default 80: two workgroups per compute unit) run fully in LDS, larger ones use panel mode so that
 *                       several workgroups share a compute unit
 *   "small_threads", "tp_threads"  workgroup size of the single-workgroup front kernel (256 / 512 / 1024) on
 *                       latency-bound levels (default 1024) and on throughput levels (default 512)
 *   "schur_first"       small fronts in panel mode with at least this many update blocks store the Schur product into their update
 *                       columns first and add the factor blocks / children's update blocks afterwards (no zero fill, no atomics
 *                       for the product); default 40, 0 = never
 *   "syrk_small_tiles"  wide trailing updates of fewer 64x64 tiles than this use 32x32 tiles (four times the workgroups, a quarter of
 *                       the K loop each); default 320 = a quarter of a round of workgroups, 0 = never
 *   "syrk_pair_tiles", "syrk_group"  multi-workgroup fronts on levels whose first wide update has at least syrk_pair_tiles 64x64 tiles
 *                       (default 2048; 0 = never) close every GROUP of syrk_group outer blocks (default 3; 2 .. 8) with ONE wide update of
 *                       K = the group's columns instead of one per 128 columns: the far part of the trailing matrix is read and written once
 *                       per group; inside a group only the next block's own columns are updated (left-looking, K = the group so far)
 *   "syrk_xcd_order"    wide trailing updates of at least this many 64x64 tiles use the XCD-aware tile order (default 512 = one
 *                       round of workgroups; 0 = never, 1 = always)
 *   "batch_extend"      1 (default): april_graph_cholesky on a graph that only GREW since the last plan keeps the plan -- the
 *                       appended poses become tail fronts, every front is re-factorised (batch semantics) -- instead of a new
 *                       ordering + symbolic analysis per call; once more than "extend_tail_fronts" (default 3) x 24 poses have been
 *                       appended (the unit is fixed at 24 poses whatever "tail_poses" -- default 28 -- says), or when the topology stops
 *                       changing, a full re-plan follows.  0 = re-plan on every topology change
 *   "pin_last"          k > 0: the k newest poses are kept out of the nested dissection and form the root front ("recent
 *                       poses last", cf. aprilsam.c:1021-1098); default 0, measured effect in profiles/r02_inc_hist.json
 *   "speculate_factors" 1 (default): a warm april_graph_cholesky call on an unchanged graph launches the step on the packed factor
 *                       copies first and reads every factor object (z / W edited in place?) while the GPU works; an edit voids that
 *                       run and the call starts over (stats.reserved1 = 1).  0 = read the factor objects before launching
 *   "warm_up"           1 (default): the first april_graph_cholesky_param_init of a process initialises what the HIP runtime sets up
 *                       lazily -- two streams (8-20 ms each: a graph owns one), the copy engines' queues (7 ms per direction), the code object (2 ms), the graph
 *                       machinery (8 ms) -- so that those milliseconds do not land inside the first solver calls (or, for the first
 *                       device-to-host copy, inside an incremental step a thousand steps into a run).  Without a device it does
 *                       nothing.  0 (set APRILSAM_AMD_WARM_UP=0 in the environment: it is read before the first param exists) =
 *                       everything stays lazy.  Streams are recycled per device slot either way (a released graph parks its stream)
 *   "inc_fast"          0 = every incremental step re-plans (default 1: frozen base plan + dirty root paths)
 *   "inc_multi"         0 = incremental steps launch their fronts / back substitution level by level (default 1: one multi-level
 *                       launch per direction, fronts synchronised by dependency flags)
 *   "inc_one"           1 (default, needs inc_multi): an incremental step that regenerates at most "inc_one_up" (3) fronts and walks
 *                       at most "inc_one_dn" (4) runs as ONE launch of one workgroup of "inc_one_threads" (512) threads -- patches,
 *                       linearisation, fronts, back substitution, state update; "inc_one_spin" (1): its completion is a word in
 *                       pinned host memory the host spins on (0: hipStreamSynchronize)
 *   "inc_tail"          1 (default, needs inc_multi): steps whose new factors touch only the last 8 poses of the last tail front
 *                       re-factorise that front's trailing columns alone (the front keeps the shape of a full one through phantom
 *                       rows); 0 = every dirty front is re-assembled and re-factorised in full
 *   "inc_inline"        1 (default): a small step's patches (<= 24 ranges, <= 2 KiB) travel in the kernel arguments; 0 = always read
 *                       from pinned host memory by the kernel
 *   "inc_update"        1 (default, needs inc_multi): a new factor between an old pose and a recent one UPDATES the factor of every front
 *                       on the old pose's root path (three vectors per factor travelling up the assembly tree; the last tail front is
 *                       still re-factorised) instead of re-assembling and re-factorising those fronts; stats.inc_fronts_updated counts
 *                       them.  0 = round-3 behaviour
 *   "inc_tail_solve"    1 (default, needs inc_tail): when every pose the step's walk visits lies among the last 8 poses, the back
 *                       substitution and the state update run inside the same single-workgroup launch, on the trailing columns alone
 *   "inc_lazy_states"   1 (default): an incremental step whose walk is partial compares only the node objects it reads (the poses of its new
 *                       factors, the visited poses, the new poses) with the library's state mirrors; 0 = every node object on every step
 *   "inc_replan_tall"   1 (default): when a front of a plan made of single-workgroup fronts only has collected so many loop-closure rows that
 *                       it no longer fits the LDS, the step re-plans; 0 = it takes the multi-workgroup path from then on
 *   "tail_poses"        own poses per tail front of the incremental path (default 28, at least 8)
 *   "persist"           1 (default): the top levels of the elimination tree -- as many as hold at most "persist_max_fronts"
 *                       (default 240) single-workgroup fronts -- run as ONE launch per sweep, fronts synchronised by
 *                       per-front dependency flags that carry the step number and are never reset (round 5 found a release in these
 *                       launches that did not wait for its L2 write-back -- a wrong result about once in 10^4 solves of chain-like graphs,
 *                       profiles/r05_flag_soak.txt; the soak of the corrected build is profiles/r06_flag_soak.txt);
 *                       0 = one launch per level, no flags -- the conservative setting, M3500 then
 *                       costs about a quarter more per iteration (what the multi-level launches bought when they were introduced: 0.366 -> 0.294 ms)
 *   "blk_backsolve"     1 (default): multi-workgroup fronts are back-substituted 128 columns at a time by a chain
 *                       workgroup + helper workgroups, with the inverse diagonal blocks the factorisation left behind; 0 = one
 *                       workgroup per front, 32 columns at a time
 *   "wave_backsolve"    1 (default): fronts whose L panel fits LDS (multi-level launch, latency-bound levels, incremental
 *                       steps) are back-substituted column-per-lane -- one in-register chain per 64 columns; 0 = the
 *                       per-32-column-block kernel everywhere
 *   "linearize_staged_min"  graphs with at least this many factors (default 32768) write the J^T W J blocks out through
 *                       LDS with coalesced stores; smaller ones store directly (one latency chain less)
 *   "mem_cap_mb"        > 0: any single device buffer above this size is refused as if the device were out of memory
 *                       (error -11); 0 = off (default).  For testing the failure path
 *   "pool_guard"        debug, > 0: every frontal array of a plan is followed by a guard band of this many doubles (rounded up to 32), filled
 *                       with NaN when the plan is uploaded and checked after every synchronised step: a kernel that wrote into one ends the
 *                       call with error -16, a kernel that READ from one and used the value turns the results into NaN -- instead of a fault
 *                       that depends on where the allocation ends.  0 = off (default)
 *   "amalg", "amalg_max"  1: separator fronts of the nested dissection take in child separators where a model of the critical path (hand-over per
 *                       front against pivot chain per column) says so, up to amalg_max own poses (default 64) -- fewer dependent levels, more flops.
 *                       Measured on M3500 in round 6: no gain (profiles/r06_experiments_not_kept.txt); 0 = off (default)
 *   "pool_poison"       debug, 1: before every step the UPDATE block of every front the step (re)factorises -- what its parent reads -- and the
 *                       solution at its own positions -- what its children read -- are filled with NaN.  A dependency wait of a multi-level
 *                       launch that passes early then produces NaN / "not positive definite" with certainty instead of the previous step's
 *                       numbers (which are the right ones whenever the previous step solved the same system: the mask that hid round 5's
 *                       release defect from everything but a soak).  Costs one extra pass over the fronts.  0 = off (default)
 *   "skip_flag_waits"   debug, 1: the fronts of the batch path's multi-level factorisation launch do NOT wait for their children -- the negative
 *                       control of "pool_poison" (the result must then come back NaN / not positive definite).  0 = off (default) */
int aprilsam_amd_set_option(const char *name, double value);
/* debug, with option "pool_guard" on and after a step on this param: points the guard check at a band inside a live frontal array; returns
 * -16 (the check works), -1 when the param has no guarded plan.  The param's cached plan is dropped, as after any failure */
int aprilsam_amd_debug_guard_selftest(const april_graph_cholesky_param_t *param);
/* current value of an option (after the environment and any set_option call): 0, or -1 for an unknown name */
int aprilsam_amd_get_option(const char *name, double *value);

/* ---- device-resident benchmark/driver API: states stay in HBM between iterations -------------
 * aprilsam_amd_batch_resident() runs `iters` batch Gauss-Newton iterations back to back without
 * touching host node objects in between (states/l_points live in HBM), then writes the final states
 * back into the graph.  chi2_out (may be NULL) receives iters+1 values: chi^2 before the first
 * iteration and after each one.  ms_out (may be NULL) receives `iters` HIP-event durations of
 * the device work per iteration.  Returns 0 on success, <0 on failure (no device, not SPD, ...). */
int aprilsam_amd_batch_resident(april_graph_t *graph, april_graph_cholesky_param_t *param, int iters,
                                double *chi2_out, double *ms_out);

/* The same, in pieces (what bench.py times): begin = pack + plan + upload (states now resident in HBM);
 * steps(n, mode) enqueues n Gauss-Newton iterations on the solver's HIP stream — mode 0 asynchronously
 * (hipGraph replay), mode 1 with every kernel launch bracketed by a HIP event pair on that stream and a
 * synchronisation per iteration; sync waits and returns -2 if a pivot was not positive; chi2 evaluates
 * chi^2 of the resident states; end leaves the node objects as the same number of april_graph_cholesky calls would
 * (aprilsam.c:131-135, 311-315): state = the final state, l_point = the point the LAST step was linearised at,
 * delta_X = that step's dx -- the factorisation of that step is kept, so april_graph_cholesky_inc may follow. */
int    aprilsam_amd_resident_begin(april_graph_t *graph, april_graph_cholesky_param_t *param);
int    aprilsam_amd_resident_steps(april_graph_t *graph, april_graph_cholesky_param_t *param, int n, int mode);
int    aprilsam_amd_resident_sync(april_graph_t *graph, april_graph_cholesky_param_t *param);
double aprilsam_amd_resident_chi2(april_graph_t *graph);
int    aprilsam_amd_resident_end(april_graph_t *graph, april_graph_cholesky_param_t *param);
/* Per-kernel totals of the mode-1 passes since resident_begin (ms, launches) and the ALGORITHMIC flops /
 * bytes one iteration asks of each kernel (SURVEY.md §8(d) conventions).  Arrays of 16; returns the
 * number of kernels filled; names[k] points at static strings. */
int aprilsam_amd_kernel_profile(const april_graph_cholesky_param_t *param, double *ms, long long *calls,
                                double *flops, double *bytes, const char **names);
/* The same passes per LEVEL of the assembly tree: out6[6 * l + ...] = {ms factorisation (assembly included), ms back substitution,
 * fronts, fronts on the multi-workgroup path, widest own part (scalar columns), sum c_j^2 flops of the level}; a multi-level
 * launch is booked on its first level.  Returns the number of levels (fills at most cap_levels). */
int aprilsam_amd_level_profile(const april_graph_cholesky_param_t *param, double *out6, int cap_levels);

/* ---- multi-GPU: nested-dissection subtree sharding, one process per GPU -------
 * The reference has no counterpart (it is sequential); a C host drives a sharded solve through the same graph / param
 * objects it hands to april_graph_cholesky (aprilsam.h:268-281):
 *
 *     aprilsam_amd_shard_begin(graph, param, rank, world);            every rank, same graph: identical plans, fronts split
 *                                                                     over the ranks by proportional mapping of the assembly
 *                                                                     tree; THIS rank allocates only the fronts it owns plus
 *                                                                     "ghost" update blocks of children that live elsewhere
 *     aprilsam_amd_shard_comm_unique_id(id)  on rank 0, id sent to the other ranks by any means (file, socket, MPI, ...)
 *     aprilsam_amd_shard_comm_init_rccl(param, id);                   RCCL communicator over xGMI on the library's device
 *       -- or aprilsam_amd_shard_comm_init_host(param, &callbacks);   the caller moves pinned host buffers itself
 *     aprilsam_amd_shard_iterate(graph, param, n);                    n Gauss-Newton iterations; the exchange happens inside:
 *                                                                     per level, the Schur update of every front whose parent
 *                                                                     lives on another rank (packed lower trapezoid, point to
 *                                                                     point); on the way down the solved x of the "top"
 *                                                                     fronts (broadcast).  With RCCL everything is enqueued
 *                                                                     on the solver's HIP stream: no host synchronisation
 *                                                                     between a level's kernels and its transfers.
 *     aprilsam_amd_shard_chi2(graph, param);                          chi^2 of the whole graph (partial sums added)
 *     aprilsam_amd_shard_gather_states(graph, param);                 every rank ends up with ALL states / l_points / dx,
 *                                                                     in HBM and in the node objects
 *     aprilsam_amd_shard_end(param);
 *
 * No all-reduce on the data path; world = 1 needs no transport.
 *   shard_info what: 0 -> {levels, fronts, nodes, pool doubles of this rank, pool doubles of the whole plan};
 *                    1 -> transfers {level, front, src, dst, (unused), packed count in doubles};
 *                    2 -> broadcasts {level, front, owner, first position, blocks}; 3 -> owner rank per front;
 *                    4 -> modelled critical path in sum c_j^2 flops {whole factorisation, heaviest root path through the fronts that
 *                         span several ranks (each runs on one owner), the busiest rank's own subtrees, all such fronts together}:
 *                         speed-up bound = [0] / ([1] + [2])
 * Return codes: 0 ok; -1 bad arguments / no shard_begin; -2 non-positive pivot; -4 foreign factor types; -5 librccl.so not
 * loadable; -6 communication error (RCCL or host callback; message on stderr); -7 no transport attached. */
typedef struct aprilsam_amd_host_comm {
    void *user;                                                           /* passed back to every callback */
    int (*send)(void *user, const double *buf, long long count, int dst);           /* blocking; 0 = ok */
    int (*recv)(void *user, double *buf, long long count, int src);
    int (*bcast)(void *user, double *buf, long long count, int root);               /* in place */
    int (*allreduce_sum)(void *user, double *buf, long long count);                 /* in place */
} aprilsam_amd_host_comm_t;
int       aprilsam_amd_shard_begin(april_graph_t *graph, april_graph_cholesky_param_t *param, int rank, int world);
long long aprilsam_amd_shard_info(const april_graph_cholesky_param_t *param, int what, long long *out, long long cap);
int       aprilsam_amd_shard_comm_unique_id(char *out128);
int       aprilsam_amd_shard_comm_init_rccl(april_graph_cholesky_param_t *param, const char *id128);
int       aprilsam_amd_shard_comm_init_host(april_graph_cholesky_param_t *param, const aprilsam_amd_host_comm_t *callbacks);
/* the attached transport as the communication library reports it: out5 = {kind (0 none, 1 RCCL, 2 host callbacks),
 * ncclCommCount, ncclCommUserRank, ncclGetVersion code, HIP device}; rccl_path (may be NULL): the librccl file in use */
int       aprilsam_amd_shard_comm_info(const april_graph_cholesky_param_t *param, long long *out5, char *rccl_path, int cap);
int       aprilsam_amd_shard_iterate(april_graph_t *graph, april_graph_cholesky_param_t *param, int n);
int       aprilsam_amd_shard_gather_states(april_graph_t *graph, april_graph_cholesky_param_t *param);
double    aprilsam_amd_shard_chi2(april_graph_t *graph, april_graph_cholesky_param_t *param);
void      aprilsam_amd_shard_end(april_graph_cholesky_param_t *param);

/* ---- host-logic introspection (no GPU needed): ordering + symbolic analysis of a graph -------- */
typedef struct aprilsam_amd_plan aprilsam_amd_plan_t;   /* opaque */
aprilsam_amd_plan_t *aprilsam_amd_plan_create(int n_nodes, int n_factors, const int *factor_nodes /* 2 per factor, -1 for unary */,
                                              const double *xy /* 2 per node or NULL */, int leaf_nodes);
void aprilsam_amd_plan_destroy(aprilsam_amd_plan_t *plan);
/* query: fills *out with a malloc'ed int64 array the caller frees with aprilsam_amd_free(); returns its length.
 * what: "perm" (position->node), "front_ptr" , "front_nsb", "front_nub", "front_parent", "front_level",
 *       "front_rows_ptr", "front_rows" (block positions of every front's rows: own then struct),
 *       "factor_front", "stats" (n_fronts, n_levels, max_rows, nnzL, flops) */
long long aprilsam_amd_plan_query(const aprilsam_amd_plan_t *plan, const char *what, long long **out);
/* Ownership map and exchange lists of a `world`-rank sharded run of this plan, as aprilsam_amd_shard_info reports them
 * (what: 1 transfers x6, 2 broadcasts x5, 3 owner per front, 4 modelled critical path x4); host logic only. */
long long aprilsam_amd_shard_plan(const aprilsam_amd_plan_t *plan, int world, int what, long long *out, long long cap);
void aprilsam_amd_free(void *p);

/* Host logic behind the incremental path: the reference's elimination order (aprilsam.c:999-1249, restated
 * with its tie-breaking) and the block elimination tree it implies (aprilsam.c:613-657).  fb[i] < 0 marks a
 * unary factor.  out_order: position -> node; out_parent (may be NULL): node -> parent node or -1. */
int aprilsam_amd_reference_order(int n_nodes, int n_factors, const int *fa, const int *fb, int *out_order, int *out_parent);

/* test handle on the same bookkeeping model, stepped explicitly (see tests/test_refmodel.py) */
void *aprilsam_amd_refmodel_create(void);
void  aprilsam_amd_refmodel_destroy(void *m);
void  aprilsam_amd_refmodel_batch(void *m, int n_nodes, int n_factors, const int *fa, const int *fb);
int   aprilsam_amd_refmodel_inc_begin(void *m, int n_nodes, int n_factors, const int *fa, const int *fb);   /* -> naffected */
int   aprilsam_amd_refmodel_solve_visit(void *m, const double *x, double dxy, double dth, int *visited);    /* -> start_over */
void  aprilsam_amd_refmodel_get(void *m, int *parent, int *changed, int *relin);
int   aprilsam_amd_refmodel_check(void *m);   /* 0: the incrementally maintained tree equals a full recomputation */

/* ---- synthetic Manhattan lattice generator ----------------------- */
/* Appends K*K xyt nodes, the in-bounds 4-direction xyt factors and the node-0 prior to `graph`.
 * Returns the number of factors added. */
int aprilsam_amd_make_lattice(april_graph_t *graph, int K);
/* The same lattice as plain arrays: states[3*K*K], fa/fb[F] (fb = -1 for the prior), z[3F], W[9F] with
 * F = 2K(K-1) + 2(K-1)^2 + 1 (caller allocates).  Returns F. */
int aprilsam_amd_lattice_arrays(int K, double *states, int *fa, int *fb, double *z, double *W);
/* Bulk append N xyt nodes (state = init = truth) and F factors (fb[i] < 0: xytpos prior on fa[i]). */
void aprilsam_amd_graph_from_arrays(april_graph_t *graph, int N, const double *states, int F, const int *fa,
                                    const int *fb, const double *z, const double *W);

/* Bulk read of the node objects: state / l_point / delta_X of every node (3 doubles each, node order; any destination may
 * be NULL).  Assumes xyt nodes (3 degrees of freedom), like the two calls above. */
void aprilsam_amd_graph_node_arrays(const april_graph_t *graph, double *state, double *l_point, double *delta_X);

/* Stage-level parity exports (tests/test_gpu_stages.py): the device linearisation and the gather assembly seen in the
 * caller's node coordinates, at the graph's current states (l_point <- state first, as a batch step does).
 *   what 0: out[33 * F], per factor (J_a^T W) J_a, (J_a^T W) J_b, (J_b^T W) J_b (3 x 3 row-major each), (J_a^T W) r, (J_b^T W) r
 *           -- the products the reference forms at aprilsam.c:159-192 from april_graph_xyt.c:62-124 / april_graph_xytpos.c:63-102
 *   what 1: out[9 N^2 + 3 N]: A = sum J^T W J + tikhanov * I (dense, symmetric, row-major) then B = sum J^T W r, both in
 *           node order -- param->A (un-permuted) and param->B of aprilsam.c:159-204; N <= 2000
 * Returns 0, or < 0 (empty graph, foreign factor types, too large). */
int aprilsam_amd_debug_stage(april_graph_t *graph, april_graph_cholesky_param_t *param, int what, double *out);

/* debug (env APRILSAM_AMD_KPROF=1): 16 wall-clock stamps (100 MHz ticks) per front from the last numeric pass */
int aprilsam_amd_debug_front_times(const april_graph_cholesky_param_t *param, long long *out, int n_fronts);

const char *aprilsam_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* APRILSAM_AMD_H */

// solver.h — internal interface between the C-ABI layer (capi.cpp) and the HIP runtime (solver.hip.cpp)
#pragma once
#include <vector>
#include "../../include/aprilsam_amd.h"

namespace asam {

struct Options {
    int leaf_nodes = 16;          // nested-dissection leaf size (pose nodes)
    int deterministic = 0;        // 1: the wall-clock fallback rule aprilsam.c:557-559 is disabled (0 = the reference's behaviour)
    int use_graph = 1;            // replay the numeric phase from a captured hipGraph
    int device_timing = 0;        // record HIP events per stage (disables graph replay for that call)
    int trust_factor_cache = 0;   // 1: z/W of factors already packed are treated as immutable (skips the per-call content check)
    int small_lds_kb = 156;       // fronts whose LDS image fits run in the single-workgroup LDS kernel
    int inc_fast = 1;             // incremental steps regenerate only the dirty root paths (0: full re-plan per step)
    int inc_multi = 1;            // incremental steps: the regenerated fronts of all levels as ONE multi-level launch, the back substitution as another (0: one launch per level)
    int inc_one = 1;              // ... and a step that regenerates at most inc_one_up fronts and walks at most inc_one_dn as ONE single-workgroup launch
    int inc_one_up = 3, inc_one_dn = 4, inc_one_threads = 512, inc_one_spin = 1;
    int inc_tail = 1;             // ... and steps whose factors touch the last few poses of the last tail front alone re-factorise its trailing columns only
    int inc_inline = 1;           // ... a small step's table / factor / state patches travel in the kernel arguments instead of being read across PCIe
    int inc_tail_solve = 1;       // ... and when every pose the step's walk visits lies among those trailing columns, the back substitution and the state update happen in the same LDS window
    int inc_lazy_states = 1;      // ... and a step whose walk is partial compares only the node objects it reads (poses of its new factors, visited poses) with the state mirrors, not all of them
    int inc_replan_tall = 1;      // ... and when a front of an all-single-workgroup plan outgrows the LDS (rows collected from loop closures), the step re-plans instead of taking the multi-launch path from then on
    int inc_update = 1;           // ... and the fronts on the root path of a loop closure take a low-rank UPDATE of their factor (front_update_body) instead of being re-assembled and re-factorised
    int speculate_factors = 1;    // warm batch calls: the pass over the factor objects (edits in place) runs under the GPU's work on the packed copies; an edit voids the run
    int warm_up = 1;              // april_graph_cholesky_param_init initialises the HIP runtime's lazy parts once per process (a stream, the copy engines' queues, the code object, the graph machinery) instead of the first solver calls; 0 = leave them lazy (APRILSAM_AMD_WARM_UP=0: the option is read from the environment before the first param exists)
    int syrk_xcd_order = 512;     // wide trailing updates of at least this many tiles (one round of workgroups is 512): tile order in which every XCD works on 8 x 8 blocks of tiles (kernels.hip.h: trapezoid_tile_xcd); 0 = never
    int syrk_small_tiles = 320;   // wide trailing updates of fewer 64 x 64 tiles than this (a quarter of a round of workgroups) use 32 x 32 tiles; 0 = never.  Measured on the 100 k lattice: k_syrk_big 0.664 (never) / 0.633 (320) / 0.648 (640) / 0.676 (1280) ms -- such a launch is 27 us of start / end latencies whatever its tiles
    int syrk_pair_tiles = 2048;   // big fronts: levels whose first wide update has at least this many 64 x 64 tiles close every PAIR of outer blocks with one K = 256 update (after the first block of a pair only the next block's columns are updated); 0 = never.  Smaller updates are a launch's worth of latency whatever their K: pairs only add a launch there
    int syrk_group = 3;           // ... outer blocks per group (2: pairs, K = 256; 3: K = 384 -- measured best on the 1 M lattice: k_syrk_big 14.9 -> 13.1 (2) -> 12.75 (3) -> 12.9 ms (4, 6); inside a group the next block's columns are updated left-looking with K = the group's blocks so far)
    int schur_first = 40;         // panel-mode small fronts with at least this many update blocks: update columns assembled after the Schur product has been stored into them (0 = never; M3500's fronts stay below: on its latency path the second assembly pass costs more than the zero fill it saves)
    int small_threads = 1024;     // workgroup size of k_front_small (256 / 512 / 1024) on latency-bound levels ...
    int tp_threads = 512;         // ... and on throughput levels (>= tp_fronts fronts)
    int tp_fronts = 1000;         // levels with at least this many fronts are "throughput levels" ...
    int tp_lds_kb = 80;           // ... where only fronts up to this LDS size run fully in LDS (the rest: panel mode, more workgroups per CU).  80 KB = two such workgroups per compute unit; measured, round 6, k_front_small ms per iteration at 48 / 64 / 80 / 96 / 160: 100 k lattice 0.976 / 0.936-0.955 / 0.924-0.936 / 1.174 / 1.100, 1 M lattice 8.65 / 8.22-8.36 / 7.94-8.05 / 10.98 / 10.59
    int amalg = 0;                // 1 = separator amalgamation (symbolic.cpp: amalgamate): separator fronts take in child separators where a cost model of the critical path (hand-over per front vs pivot chain per column) says so -- fewer dependent levels on graphs the size of M3500
    int amalg_max = 64;           // ... as long as the merged front owns at most this many poses
    int pin_last = 0;             // nested dissection keeps the pin_last newest poses out of the dissection: they form the root front ("recent poses last")
    int batch_extend = 1;         // batch calls on a graph that only grew reuse the plan: appended poses become tail fronts, every front is re-factorised
    int extend_tail_fronts = 3;   // ... until the appended poses are this many tail fronts' worth, units of 24 poses (then: full re-plan).  Measured, round 4: demo --batch_update_only 1 500 poses 481 / 388 / 372 / 371 ms at 8 / 4 / 3 / 2; the incremental demo does not care (510 +- 3 %)
    int persist = 1;              // batch path: the top levels of the tree (few small fronts each) as ONE launch per sweep, fronts synchronised by dependency flags
    int persist_max_fronts = 240; // ... as many top levels as fit this many fronts
    int tail_poses = 28;          // incremental path: own poses per tail front (>= 8; measured on the M3500 demo: 24 / 28 / 32 -> 546 / 530 / 528 ms total, median 0.038 / 0.0385 / 0.040 ms)
    int blk_backsolve = 1;        // wide multi-workgroup fronts: back substitution 128 columns at a time by a chain workgroup + helpers (0: k_backsolve_gemv + k_backsolve_t)
    int wave_backsolve = 1;       // multi-level back substitution: column-per-lane kernel (0: the per-32-column-block kernel)
    int linearize_staged_min = 32768; // factors per launch from which k_linearize writes its results out through LDS (coalesced stores)
    int mem_cap_mb = 0;           // > 0: refuse any single device buffer above this size with ERR_OOM (tests: the out-of-memory path)
    int pool_guard = 0;           // debug: > 0 = every frontal array of a plan is followed by a guard band of this many doubles, NaN-filled at plan upload and checked after every synchronised step (ERR_GUARD); a stray read that is used poisons the result
    int pool_poison = 0;          // debug: 1 = before every step, the update block of every front the step (re)factorises and x at its own positions are filled with NaN: a dependency wait of a multi-level launch that passes early yields NaN instead of the previous step's numbers
    int skip_flag_waits = 0;      // debug, negative control of pool_poison: 1 = the fronts of the batch path's multi-level factorisation launch do NOT wait for their children
    int panel_mode = 1;           // fronts too large for LDS whose own columns fit run in k_front_small's panel mode
};
extern Options g_opt;

void batch_step(april_graph_t *g, april_graph_cholesky_param_t *param);
void inc_step(april_graph_t *g, april_graph_cholesky_param_t *param);
void inc_solve_only(april_graph_t *g, april_graph_cholesky_param_t *param);
double graph_chi2(april_graph_t *g);
int batch_resident(april_graph_t *g, april_graph_cholesky_param_t *param, int iters, double *chi2_out, double *ms_out);
int resident_begin(april_graph_t *g, april_graph_cholesky_param_t *param);
int resident_steps(april_graph_t *g, april_graph_cholesky_param_t *param, int n, int mode);
int resident_sync(april_graph_t *g, april_graph_cholesky_param_t *param);
double resident_chi2(april_graph_t *g);
int resident_end(april_graph_t *g, april_graph_cholesky_param_t *param);
int level_profile(const april_graph_cholesky_param_t *param, double *out, int cap_levels);
int kernel_profile(const april_graph_cholesky_param_t *param, double *ms, long long *calls, double *flops, double *bytes, const char **names);
void drop_context(const april_graph_cholesky_param_t *p);
void drop_graph_pack(const april_graph_t *g);
bool get_stats(const april_graph_cholesky_param_t *p, aprilsam_amd_stats_t *out);
int shard_begin(april_graph_t *g, april_graph_cholesky_param_t *param, int rank, int world);
long long shard_info(const april_graph_cholesky_param_t *param, int what, long long *out, long long cap);
int shard_comm_unique_id(char *out128);
int shard_comm_init_rccl(const april_graph_cholesky_param_t *param, const char *id128);
int shard_comm_init_host(const april_graph_cholesky_param_t *param, const aprilsam_amd_host_comm_t *cb);
int shard_comm_info(const april_graph_cholesky_param_t *param, long long *out, char *path, int cap);
int shard_iterate(april_graph_t *g, april_graph_cholesky_param_t *param, int n);
int shard_gather_states(april_graph_t *g, april_graph_cholesky_param_t *param);
double shard_chi2(april_graph_t *g, april_graph_cholesky_param_t *param);
void shard_end(const april_graph_cholesky_param_t *param);
int debug_guard_selftest(const april_graph_cholesky_param_t *param);
int debug_stage(april_graph_t *g, april_graph_cholesky_param_t *param, int what, double *out);
int debug_front_times(const april_graph_cholesky_param_t *param, long long *out, int n_fronts);
int api_device_count();
int selftest();
struct Plan;
void shard_map(const Plan &P, int world, std::vector<int> &owner, std::vector<char> &top, std::vector<long long> &xfer, std::vector<long long> &bcast);
std::vector<long long> shard_critical_path(const Plan &P, int world, const std::vector<int> &owner, const std::vector<char> &top);
int api_set_device(int d);
int api_param_set_device(const april_graph_cholesky_param_t *param, int slot);
int api_param_get_device(const april_graph_cholesky_param_t *param);
void unbind_param(const april_graph_cholesky_param_t *param);
void warm_up() noexcept;          // once per process, from april_graph_cholesky_param_init: see solver.hip.cpp
int api_set_option(const char *name, double v);
int api_get_option(const char *name, double *v);

}  // namespace asam

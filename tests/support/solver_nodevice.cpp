// solver_nodevice.cpp -- TEST INFRASTRUCTURE for the sanitizer build of the HOST sources (tools/sanitize_host.sh):
// ordering.cpp, symbolic.cpp, refmodel.cpp, graph_io.cpp, host_objects.cpp, errors.cpp and capi.cpp are compiled by g++ with
// -fsanitize=address,undefined and linked with this file in place of solver.hip.cpp (the HIP translation unit, which hipcc's
// sanitizer runtime cannot serve inside a shared library loaded by an uninstrumented python).  Every entry point that needs the
// GPU does here what the real library does on a box without a device: it computes nothing, records error -14 and says so.  The
// host-side logic under test (planner, reference-order model, .graph files, object constructors, C-ABI glue) is the product's own.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../aprilsam_amd/csrc/errors.h"
#include "../../aprilsam_amd/csrc/plan.h"
#include "../../aprilsam_amd/csrc/solver.h"

namespace asam {

Options g_opt;
static int no_device(const char *what) {
    const std::string msg = std::string(what) + ": sanitizer build of the host sources, no HIP translation unit (there is NO CPU fallback: nothing was computed)";
    fprintf(stderr, "aprilsam_amd: ERROR -14: %s\n", msg.c_str());
    set_last_error(ERR_NO_DEVICE, msg);
    return ERR_NO_DEVICE;
}
void batch_step(april_graph_t *, april_graph_cholesky_param_t *) { no_device("april_graph_cholesky"); }
void inc_step(april_graph_t *, april_graph_cholesky_param_t *) { no_device("april_graph_cholesky_inc"); }
void inc_solve_only(april_graph_t *, april_graph_cholesky_param_t *) { no_device("april_graph_cholesky_inc_solver"); }
double graph_chi2(april_graph_t *) { no_device("april_graph_chi2"); return NAN; }
int batch_resident(april_graph_t *, april_graph_cholesky_param_t *, int, double *, double *) { return no_device("batch_resident"); }
int resident_begin(april_graph_t *, april_graph_cholesky_param_t *) { return no_device("resident_begin"); }
int resident_steps(april_graph_t *, april_graph_cholesky_param_t *, int, int) { return no_device("resident_steps"); }
int resident_sync(april_graph_t *, april_graph_cholesky_param_t *) { return no_device("resident_sync"); }
double resident_chi2(april_graph_t *) { no_device("resident_chi2"); return NAN; }
int resident_end(april_graph_t *, april_graph_cholesky_param_t *) { return no_device("resident_end"); }
int level_profile(const april_graph_cholesky_param_t *, double *, int) { return -1; }
int kernel_profile(const april_graph_cholesky_param_t *, double *, long long *, double *, double *, const char **) { return -1; }
void drop_context(const april_graph_cholesky_param_t *) {}
void drop_graph_pack(const april_graph_t *) {}
bool get_stats(const april_graph_cholesky_param_t *, aprilsam_amd_stats_t *out) { if (out) { memset(out, 0, sizeof(*out)); out->error_code = ERR_NO_DEVICE; } return out != nullptr; }
int shard_begin(april_graph_t *, april_graph_cholesky_param_t *, int, int) { return no_device("shard_begin"); }
long long shard_info(const april_graph_cholesky_param_t *, int, long long *, long long) { return -1; }
int shard_comm_unique_id(char *) { return no_device("shard_comm_unique_id"); }
int shard_comm_init_rccl(const april_graph_cholesky_param_t *, const char *) { return no_device("shard_comm_init_rccl"); }
int shard_comm_init_host(const april_graph_cholesky_param_t *, const aprilsam_amd_host_comm_t *) { return no_device("shard_comm_init_host"); }
int shard_comm_info(const april_graph_cholesky_param_t *, long long *, char *, int) { return -1; }
int shard_iterate(april_graph_t *, april_graph_cholesky_param_t *, int) { return no_device("shard_iterate"); }
int shard_gather_states(april_graph_t *, april_graph_cholesky_param_t *) { return no_device("shard_gather_states"); }
double shard_chi2(april_graph_t *, april_graph_cholesky_param_t *) { no_device("shard_chi2"); return NAN; }
void shard_end(const april_graph_cholesky_param_t *) {}
int debug_stage(april_graph_t *, april_graph_cholesky_param_t *, int, double *) { return no_device("debug_stage"); }
int debug_guard_selftest(const april_graph_cholesky_param_t *) { return -1; }
int debug_front_times(const april_graph_cholesky_param_t *, long long *, int) { return -1; }
int api_device_count() { return 0; }
int selftest() { return -1; }                 // (the index arithmetic it checks lives in the HIP translation unit)
void shard_map(const Plan &, int, std::vector<int> &, std::vector<char> &, std::vector<long long> &, std::vector<long long> &) {}
std::vector<long long> shard_critical_path(const Plan &, int, const std::vector<int> &, const std::vector<char> &) { return {}; }
int api_set_device(int) { return -1; }
int api_param_set_device(const april_graph_cholesky_param_t *, int) { return -1; }
int api_param_get_device(const april_graph_cholesky_param_t *) { return -1; }
void unbind_param(const april_graph_cholesky_param_t *) {}
void warm_up() noexcept {}
int api_set_option(const char *name, double v) {
    if (name && !strcmp(name, "leaf_nodes")) { g_opt.leaf_nodes = (int)v; return 0; }
    if (name && !strcmp(name, "pin_last")) { g_opt.pin_last = (int)v; return 0; }
    if (name && !strcmp(name, "amalg")) { g_opt.amalg = (int)v; return 0; }
    if (name && !strcmp(name, "amalg_max")) { g_opt.amalg_max = (int)v; return 0; }
    return -1;
}
int api_get_option(const char *name, double *v) {
    if (name && v && !strcmp(name, "leaf_nodes")) { *v = g_opt.leaf_nodes; return 0; }
    if (name && v && !strcmp(name, "pin_last")) { *v = g_opt.pin_last; return 0; }
    if (name && v && !strcmp(name, "amalg")) { *v = g_opt.amalg; return 0; }
    if (name && v && !strcmp(name, "amalg_max")) { *v = g_opt.amalg_max; return 0; }
    return -1;
}

}  // namespace asam

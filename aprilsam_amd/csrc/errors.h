// errors.h — failure path of libaprilsam_amd.so.  The reference's entry points are `void` and crash on bad input
// (assert / NULL dereference on a non-SPD matrix, aprilsam.c:90-91,380-385 silent returns); a
// library that drives a GPU must not take the caller's process down with it.  Everything below the C-ABI reports a
// failure by throwing SolverError; the entry points catch it, leave the caller's node states untouched, print one line
// on stderr, record code + message (aprilsam_amd_last_error, aprilsam_amd_stats_t::error_code) and drop the param's
// cached plan so that the next call starts from a clean slate.  "No HIP device visible" is reported the same way
// (ERR_NO_DEVICE, on every call): there is no CPU fallback to fall back to, the message says that nothing was computed.
#pragma once
#include <string>

namespace asam {

enum {
    ERR_NONE = 0,
    ERR_NOT_SPD = -2,          // a pivot was not positive (reported through stats.not_spd as before)
    ERR_DEP_TIMEOUT = -9,      // a multi-level launch gave up waiting for a dependency flag
    ERR_HIP = -10,             // a HIP runtime call failed
    ERR_OOM = -11,             // device or pinned-host memory exhausted (or the mem_cap_mb option's limit)
    ERR_UNSUPPORTED = -12,     // node type / factor arity / front size this build does not handle
    ERR_BAD_GRAPH = -13,       // malformed input: node index out of range, factor connecting a node to itself
    ERR_NO_DEVICE = -14,       // no HIP device visible (there is no CPU fallback: the call computed nothing)
    ERR_INTERNAL = -15,        // inconsistency in the planner (a bug, not an input problem)
    ERR_GUARD = -16,           // debug option pool_guard: a kernel wrote into a guard band behind a frontal array
};

struct SolverError {
    int code;
    std::string msg;
};

[[noreturn]] void fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// last failure of any entry point in this process (thread-safe); returns the code, 0 when there was none
void set_last_error(int code, const std::string &msg);
int get_last_error(char *msg, int cap);
void clear_last_error();

}  // namespace asam

// errors.cpp — see errors.h
#include "errors.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

namespace asam {

void fail(int code, const char *fmt, ...) {
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw SolverError{ code, buf };
}

static std::mutex g_err_mu;
static int g_err_code = 0;
static std::string g_err_msg;

void set_last_error(int code, const std::string &msg) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err_code = code; g_err_msg = msg;
}
int get_last_error(char *msg, int cap) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    if (msg && cap > 0) { strncpy(msg, g_err_msg.c_str(), (size_t)cap - 1); msg[cap - 1] = 0; }
    return g_err_code;
}
void clear_last_error() {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err_code = 0; g_err_msg.clear();
}

}  // namespace asam

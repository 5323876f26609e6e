// Error reporting for the C-ABI: functions return an int (0 = ok, negative = errno style) and leave
// a human readable message behind; nothing is ever thrown across the boundary.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <string>
#include <unordered_map>

#include "ddspp_common.h"

static thread_local char g_last_error[512] = "";

// Tuning options (launch geometry, A/B route switches: DESIGN.md section 11).  A library does not read the
// environment on every call: an option is looked up in the environment ONCE (its DDSPP_* variable), cached, and can
// be set programmatically; ddspp_reload_options drops the cache (tests and A/B tools change variables in-process).
static std::mutex g_opt_mutex;
static std::unordered_map<std::string, int> g_opt_cache;

extern "C" {

void ddspp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

const char* ddspp_last_error(void) { return g_last_error; }

int ddspp_option(const char* name, int dflt) {
    std::lock_guard<std::mutex> lk(g_opt_mutex);
    auto it = g_opt_cache.find(name);
    if (it != g_opt_cache.end()) return it->second;
    const char* s = getenv(name);
    const int v = (s && *s) ? atoi(s) : dflt;
    g_opt_cache.emplace(name, v);
    return v;
}

int ddspp_set_option(const char* name, int value) {
    if (!name || !*name) return DDSPP_EINVAL;
    std::lock_guard<std::mutex> lk(g_opt_mutex);
    g_opt_cache[name] = value;
    return DDSPP_OK;
}

void ddspp_reload_options(void) {
    std::lock_guard<std::mutex> lk(g_opt_mutex);
    g_opt_cache.clear();
}

int ddspp_version(void) { return 200; }

const char* ddspp_target_arch(void) { return "gfx950"; }

}  // extern "C"

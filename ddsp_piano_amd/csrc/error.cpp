// Error reporting for the C-ABI: functions return an int (0 = ok, negative = errno style) and leave
// a human readable message behind; nothing is ever thrown across the boundary.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
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

// The launch path does not take that mutex or build a std::string (round 6): the library's own call sites pass string
// LITERALS, so their ADDRESS identifies the option.  ddspp_option_literal keeps (generation, value) of every call site's literal in
// a fixed open-addressed table of atomics; ddspp_set_option / ddspp_reload_options bump the generation, and a reader that
// finds an older generation in its slot refreshes it through the string-keyed map above (the only place that locks).
namespace {
constexpr unsigned OPT_SLOTS = 1024;                       // > 4 x the library's call sites, a power of two
struct OptSlot {
    std::atomic<const char*> key{nullptr};
    std::atomic<unsigned long long> gen_val{0};            // generation << 32 | (unsigned)value: one store publishes both
};
OptSlot g_opt_slots[OPT_SLOTS];
std::atomic<unsigned> g_opt_generation{1};                 // slots start at generation 0: never current
}  // namespace

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

int ddspp_option_literal(const char* name, int dflt) {
    const unsigned gen = g_opt_generation.load(std::memory_order_acquire);
    const unsigned h = (unsigned)(((unsigned long long)(uintptr_t)name * 0x9E3779B97F4A7C15ull) >> 40);
    for (unsigned probe = 0; probe < OPT_SLOTS; ++probe) {
        OptSlot& s = g_opt_slots[(h + probe) & (OPT_SLOTS - 1)];
        const char* k = s.key.load(std::memory_order_acquire);
        if (k == nullptr) {                                 // a free slot: claim it for this literal
            const char* expected = nullptr;
            if (s.key.compare_exchange_strong(expected, name, std::memory_order_acq_rel)) k = name;
            else k = expected;                              // somebody else took it (maybe for the same literal)
        }
        if (k != name) continue;
        const unsigned long long gv = s.gen_val.load(std::memory_order_acquire);
        if ((unsigned)(gv >> 32) == gen) return (int)(unsigned)gv;
        const int v = ddspp_option(name, dflt);             // first use, or an option was set / reloaded since
        s.gen_val.store(((unsigned long long)gen << 32) | (unsigned)v, std::memory_order_release);
        return v;
    }
    return ddspp_option(name, dflt);                        // table full (never with the library's own call sites)
}

int ddspp_set_option(const char* name, int value) {
    if (!name || !*name) return DDSPP_EINVAL;
    std::lock_guard<std::mutex> lk(g_opt_mutex);
    g_opt_cache[name] = value;
    g_opt_generation.fetch_add(1, std::memory_order_acq_rel);
    return DDSPP_OK;
}

void ddspp_reload_options(void) {
    std::lock_guard<std::mutex> lk(g_opt_mutex);
    g_opt_cache.clear();
    g_opt_generation.fetch_add(1, std::memory_order_acq_rel);
}

int ddspp_version(void) { return 200; }

const char* ddspp_target_arch(void) { return "gfx950"; }

}  // extern "C"

// Error reporting for the C-ABI: functions return an int (0 = ok, negative = errno style) and leave
// a human readable message behind; nothing is ever thrown across the boundary.
#include <stdarg.h>
#include <stdio.h>

#include "ddspp_common.h"

static thread_local char g_last_error[512] = "";

extern "C" {

void ddspp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

const char* ddspp_last_error(void) { return g_last_error; }

int ddspp_version(void) { return 100; }

const char* ddspp_target_arch(void) { return "gfx950"; }

}  // extern "C"

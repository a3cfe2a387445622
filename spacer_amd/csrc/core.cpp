// Error plumbing + version for libspacer_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/spacer_hip.h"

static thread_local char g_err[512] = "";

void spacer_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* spacer_last_error(void) { return g_err; }
extern "C" int spacer_version(void) { return 107; }   // 1.07: round 6 (spacer_gather_f32 / spacer_scatter_f32 / spacer_eos_schedule: EOS-trimmed scoring; spacer_attn_decode_shared_rows)

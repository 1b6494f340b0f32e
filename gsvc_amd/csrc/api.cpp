// gsvc_amd/csrc/api.cpp — error slot and version of libgsvc_hip.so.
#include "common.h"

#include <cstring>

namespace gsvc {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace gsvc

extern "C" const char *gsvc_last_error(void) { return gsvc::g_err; }

extern "C" const char *gsvc_version(void) { return "gsvc_hip 0.1.0 gfx950"; }

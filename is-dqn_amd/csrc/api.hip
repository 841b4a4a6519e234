// Library-level entry points of the C ABI (include/isdqn_hip.h).
#include <stdarg.h>

#include "common.h"

namespace isdqn {
static thread_local char g_last_error[512] = "";
void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}
}  // namespace isdqn

extern "C" const char* isdqn_version(void) { return "isdqn_hip 0.1 (gfx950)"; }
extern "C" const char* isdqn_last_error(void) { return isdqn::g_last_error; }

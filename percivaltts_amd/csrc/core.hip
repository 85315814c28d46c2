// Error reporting and identification of libpercival_hip.so.
#include "common.h"
#include <cstring>

namespace ptts {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ptts

extern "C" const char* ptts_version(void) { return "percival_hip 0.1.0 (round 1)"; }
extern "C" const char* ptts_device_arch(void) { return "gfx950"; }
extern "C" const char* ptts_last_error(void) { return ptts::g_err; }

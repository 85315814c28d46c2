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
static int g_deterministic = 0;
bool deterministic() { return g_deterministic != 0; }
}  // namespace ptts

extern "C" const char* ptts_version(void) { return "percival_hip 0.2.0 (round 2)"; }
extern "C" int ptts_set_deterministic(int on) { const int old = ptts::g_deterministic; ptts::g_deterministic = on ? 1 : 0; return old; }
extern "C" int ptts_get_deterministic(void) { return ptts::g_deterministic; }
extern "C" const char* ptts_device_arch(void) { return "gfx950"; }
extern "C" const char* ptts_last_error(void) { return ptts::g_err; }

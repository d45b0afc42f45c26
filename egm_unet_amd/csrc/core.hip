// Library-wide state: version, thread-local error string, device check.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void egm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int egm_version(void) { return 100; }   // 0.1.0
extern "C" const char* egm_last_error(void) { return g_err; }

extern "C" int egm_device_ok(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        egm_set_error("no HIP device available");
        return 0;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        egm_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        return 0;
    }
    return 1;
}

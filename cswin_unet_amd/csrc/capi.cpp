// Error reporting and version of the C ABI (no exceptions cross the boundary; see include/cswin_hip.h).
#include <stdarg.h>
#include <string.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

void cswin_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

const char* cswin_last_error(void) { return g_err; }

int cswin_abi_version(void) { return CSWIN_ABI_VERSION; }

// 1 if a gfx950 device is visible to the HIP runtime this library is bound to
int cswin_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        cswin_set_error("no HIP device visible");
        return 0;
    }
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        cswin_set_error("cannot query HIP device");
        return 0;
    }
    if (!strstr(prop.gcnArchName, "gfx950")) {
        cswin_set_error("device is %s, this library is built for gfx950 only", prop.gcnArchName);
        return 0;
    }
    return 1;
}

}  // extern "C"

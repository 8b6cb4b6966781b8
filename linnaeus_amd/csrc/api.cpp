// Error channel and misc entry points of the C ABI (include/lnx.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/lnx.h"

static thread_local char g_err[1024] = "";

void lnx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* lnx_last_error(void) { return g_err; }
extern "C" int lnx_version(void) { return 100; }
extern "C" int lnx_device_cus(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    return prop.multiProcessorCount;
}

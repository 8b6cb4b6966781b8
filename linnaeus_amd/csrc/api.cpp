// Error channel and misc entry points of the C ABI (include/lnx.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

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

// ---- tile scheduling of the persistent kernels (common.hpp) ----
static std::atomic<int> g_cu_margin{-1};
int persistent_cus(int cus) {
    int m = g_cu_margin.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("LNX_CU_MARGIN");
        m = e ? atoi(e) : 0;
        if (m < 0) m = 0;
        g_cu_margin.store(m, std::memory_order_relaxed);
    }
    const int room = cus - m;
    return room < 8 ? 8 : room;
}
void set_cu_margin(int m) { g_cu_margin.store(m < 0 ? 0 : m, std::memory_order_relaxed); }
bool tile_sched_static() {
    const char* e = getenv("LNX_TILE_SCHED");
    return e && strcmp(e, "static") == 0;
}
int tile_slot_of(hipStream_t st) {
    // stream handle -> slot, first come first served; a 65th stream gets -1 = the static stride (no counters: a shared slot would let two
    // overlapping persistent launches skip or repeat each other's tiles)
    static std::mutex mu;
    static hipStream_t known[64];
    static int n = 0;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < n; ++i)
        if (known[i] == st) return i;
    if (n < 64) {
        known[n] = st;
        return n++;
    }
    return -1;
}
int device_cus() {
    static std::atomic<int> cached{0};  // (one device per process: one process per GPU)
    int c = cached.load(std::memory_order_relaxed);
    if (c == 0) {
        c = lnx_device_cus();
        if (c > 0) cached.store(c, std::memory_order_relaxed);
    }
    return c > 0 ? c : 0;
}
extern "C" int lnx_set_cu_margin(int cus) {
    if (cus < 0 || cus > 1024) {  // (a launch always keeps at least 8 workgroups: persistent_cus)
        lnx_set_error("lnx_set_cu_margin: %d (0..1024 compute units)", cus);
        return 1;
    }
    set_cu_margin(cus);
    return 0;
}

// Thread-local last-error string behind the C ABI (include/rvc_amd.h: rvc_last_error).
#include <stdarg.h>

#include <mutex>
#include <unordered_set>

#include "common.h"

namespace rvc {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

// common.h: LDS_WHOLE_CU.  One hipFuncSetAttribute per (device, kernel) and process: the attribute belongs to the CURRENT device's
// function object, so a process that drives several devices (a thread per device) needs it on each.
int reserve_whole_cu(const void *kernel, const char *what) {
    static std::mutex mu;
    static std::unordered_set<const void *> done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    std::lock_guard<std::mutex> g(mu);
    if (dev < 64 && done[dev].count(kernel)) return 0;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WHOLE_CU);
    if (e != hipSuccess) return fail("%s: cannot reserve %d bytes of LDS on device %d: %s", what, LDS_WHOLE_CU, dev, hipGetErrorString(e));
    if (dev < 64) done[dev].insert(kernel);   // (beyond 64 devices: set it every time -- it is cheap)
    return 0;
}

}  // namespace rvc

extern "C" int rvc_abi_version(void) { return RVC_AMD_ABI_VERSION; }
extern "C" const char *rvc_last_error(void) { return rvc::g_err; }

// Thread-local last-error string behind the C ABI (include/rvc_amd.h: rvc_last_error).
#include <stdarg.h>

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

}  // namespace rvc

extern "C" int rvc_abi_version(void) { return RVC_AMD_ABI_VERSION; }
extern "C" const char *rvc_last_error(void) { return rvc::g_err; }

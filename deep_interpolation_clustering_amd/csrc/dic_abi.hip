// Version / error reporting half of the C ABI (include/dic_hip.h).
#include <cstdarg>
#include <cstdio>
#include "dic_common.h"

namespace dic {

static thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DIC_ERR_LAUNCH;
    }
    return DIC_OK;
}

}  // namespace dic

extern "C" {

int dic_version(void) { return DIC_ABI_VERSION; }

const char* dic_status_string(int status) {
    switch (status) {
        case DIC_OK: return "DIC_OK";
        case DIC_ERR_INVALID_ARG: return "DIC_ERR_INVALID_ARG";
        case DIC_ERR_UNSUPPORTED: return "DIC_ERR_UNSUPPORTED";
        case DIC_ERR_WORKSPACE: return "DIC_ERR_WORKSPACE";
        case DIC_ERR_LAUNCH: return "DIC_ERR_LAUNCH";
        default: return "DIC_ERR_UNKNOWN";
    }
}

const char* dic_last_error_string(void) { return dic::g_last_error; }

}  // extern "C"

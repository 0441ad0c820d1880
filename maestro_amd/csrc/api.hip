// Error plumbing + version of the C ABI (include/maestro_hip.h).
#include "common.hpp"
#include "../../include/maestro_hip.h"

thread_local char mh_err_buf[512] = {0};

int mh_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(mh_err_buf, sizeof(mh_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* mh_last_error(void) { return mh_err_buf; }
extern "C" int mh_version(void) { return 1; }

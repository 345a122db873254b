// libdifformer_hip.so: version / error plumbing of the C ABI (include/difformer_hip.h).
#include <stdarg.h>
#include <stdlib.h>
#include "dif_common.h"

namespace dif {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(static_cast<int>(e), "%s: %s", what, hipGetErrorString(e));
    return 0;
}

}  // namespace dif

extern "C" int dif_version(void) { return DIF_ABI_VERSION; }
extern "C" const char* dif_last_error(void) { return dif::err_buf(); }

bool dif::exact_fp32() {
    static const bool on = [] { const char* e = getenv("DIFFORMER_EXACT_FP32"); return e && e[0] == '1'; }();
    return on;
}

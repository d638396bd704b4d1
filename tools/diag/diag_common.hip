#include "diag_common.h"

namespace sctc {
char* diag_err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}
int set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(diag_err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace sctc

extern "C" const char* sctc_diag_last_error(void) { return sctc::diag_err_buf(); }

// ABI bookkeeping: version and the thread-local error message.
#include "snerf_common.h"

namespace snerf {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace snerf

extern "C" int snerf_abi_version(void) { return SNERF_ABI_VERSION; }
extern "C" const char* snerf_last_error(void) { return snerf::error_buffer(); }

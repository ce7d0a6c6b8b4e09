// capi.hip — error plumbing shared by every entry point of libtoad_hip.so.
#include "common.h"

#include <stdarg.h>
#include <atomic>

namespace toad {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static std::atomic<long long> g_fallback_launches{0};
void note_fallback_launch() { g_fallback_launches.fetch_add(1, std::memory_order_relaxed); }

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return TOAD_OK;
}

}  // namespace toad

extern "C" int toad_abi_version(void) { return TOAD_ABI_VERSION; }
extern "C" const char *toad_last_error(void) { return toad::g_err; }
extern "C" int64_t toad_fallback_launches(void) { return (int64_t)toad::g_fallback_launches.load(std::memory_order_relaxed); }

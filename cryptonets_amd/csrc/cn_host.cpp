// Host-only pieces of libcnhip.so shared by every translation unit: the thread-local error message of the C ABI and the context lock.
#include "cn_runtime.h"
#include <cstdarg>
#include <cstdio>
#include <sched.h>

static thread_local char g_err[512] = "";
int cn_fail(int code, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}
extern "C" const char *cn_last_error(void) { return g_err; }

void CnMutex::lock() {
    for (int spins = 0;; spins++) {
        if (!s.load(std::memory_order_relaxed) && !s.exchange(1, std::memory_order_acquire)) return;
        if (spins < 4096) __builtin_ia32_pause(); else sched_yield();
    }
}
void CnMutex::unlock() { s.store(0, std::memory_order_release); }

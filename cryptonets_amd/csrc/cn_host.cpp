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

// Test-and-test-and-set with an early yield.  (A queue lock - MCS: waiters spin on their own node, hand-over in arrival order - was
// measured on the GPU box and lost clearly: 22.7 vs 16.3 ms per CryptoNets batch at 8 caller threads, 82 vs 34 ms at 64: the caller
// threads of a layer are short-lived and the box hands them fewer cores than they are, and a FIFO lock waits for exactly the one
// successor that is not running.)
void CnMutex::lock(Node &) {
    for (int spins = 0;; spins++) {
        if (!held.load(std::memory_order_relaxed) && !held.exchange(1, std::memory_order_acquire)) return;
        if (spins < 64) __builtin_ia32_pause(); else sched_yield();
    }
}
void CnMutex::unlock(Node &) { held.store(0, std::memory_order_release); }

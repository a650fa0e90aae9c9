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

static inline void spin_wait(int &spins) { if (++spins < 64) __builtin_ia32_pause(); else sched_yield(); }
void CnMutex::lock(Node &n) {
    n.next.store(nullptr, std::memory_order_relaxed);
    n.locked.store(1, std::memory_order_relaxed);
    Node *prev = tail.exchange(&n, std::memory_order_acq_rel);
    if (!prev) return;                                       // free: the lock is ours
    prev->next.store(&n, std::memory_order_release);
    for (int spins = 0; n.locked.load(std::memory_order_acquire);) spin_wait(spins);
}
void CnMutex::unlock(Node &n) {
    Node *succ = n.next.load(std::memory_order_acquire);
    if (!succ) {
        Node *expect = &n;
        if (tail.compare_exchange_strong(expect, nullptr, std::memory_order_acq_rel)) return;     // nobody waits
        for (int spins = 0; !(succ = n.next.load(std::memory_order_acquire));) spin_wait(spins);    // a waiter has swapped the tail but not linked itself yet
    }
    succ->locked.store(0, std::memory_order_release);
}

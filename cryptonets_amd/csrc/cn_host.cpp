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

// The context lock.  The reference calls the wrapper from Defaults.ThreadCount = Environment.ProcessorCount threads (HE Wrapper/Defaults.cs,
// Utils.cs:46-88) - 256 on the bench box - and every call is a few hundred nanoseconds of bookkeeping under this lock: the work is
// serial whatever the thread count, so what matters is that the waiters stay out of the holder's way.
//   * At most MAX_SPINNERS waiters contend for the lock at a time: test-and-test-and-set with an early yield - the regime measured best
//     for 1-8 caller threads (a FIFO queue lock - MCS - lost clearly on the GPU box: 22.7 vs 16.3 ms per CryptoNets batch at 8 threads, 82
//     vs 34 ms at 64: caller threads are short-lived, the box hands them fewer cores than they are, and a FIFO lock waits for exactly the
//     successor that is not running).
//   * Every further waiter SLEEPS (clock_nanosleep, 50 us doubling to 400 us) and looks again when it wakes: no wake-up call on the unlock
//     path (a futex hand-over would put a system call into every one of the ~20 000 critical sections of a batch), no run-queue
//     pressure from hundreds of yielding threads (round 2: 0.47 of the batched rate at 64 threads, worse beyond), and the cap no longer
//     has to be applied by the caller (GpuSealBfvFactory's callerThreads): an unchanged program with 256 caller threads behaves like one
//     with MAX_SPINNERS + 1.
#include <time.h>
static const int MAX_SPINNERS = 3;
void CnMutex::lock(Node &) {
    if (!held.load(std::memory_order_relaxed) && !held.exchange(1, std::memory_order_acquire)) return;
    unsigned sleep_us = 50;
    for (;;) {
        if (spinners.fetch_add(1, std::memory_order_relaxed) < MAX_SPINNERS) {
            for (int spins = 0;; spins++) {
                if (!held.load(std::memory_order_relaxed) && !held.exchange(1, std::memory_order_acquire)) { spinners.fetch_sub(1, std::memory_order_relaxed); return; }
                if (spins < 64) __builtin_ia32_pause(); else sched_yield();
            }
        }
        spinners.fetch_sub(1, std::memory_order_relaxed);
        struct timespec ts = {0, (long)sleep_us * 1000};
        clock_nanosleep(CLOCK_MONOTONIC, 0, &ts, nullptr);
        if (sleep_us < 400) sleep_us *= 2;
        if (!held.load(std::memory_order_relaxed) && !held.exchange(1, std::memory_order_acquire)) return;      // free right now: take it
    }
}
void CnMutex::unlock(Node &) { held.store(0, std::memory_order_release); }

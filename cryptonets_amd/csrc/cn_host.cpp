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
// serial whatever the thread count, so what matters is that the waiters stay out of the holder's way and that nobody oversleeps.
//   * At most MAX_SPINNERS waiters contend for the lock at a time: test-and-test-and-set with an early yield - the regime measured best
//     for 1-8 caller threads (a FIFO queue lock - MCS - lost clearly on the GPU box: 22.7 vs 16.3 ms per CryptoNets batch at 8 threads, 82
//     vs 34 ms at 64: caller threads are short-lived, the box hands them fewer cores than they are, and a FIFO lock waits for exactly the
//     successor that is not running).
//   * Every further waiter SLEEPS on a futex word.  An unlock wakes ONE sleeper, and only when nobody is spinning - i.e. when the lock
//     would otherwise go idle; while spinners exist they take the lock over and the unlock path is a plain store (a wake-up call in
//     every one of the ~20 000 critical sections of a batch would cost more than the sections).  At the end of a parallel region the
//     active threads run out of items, the spinner count drops to zero and the sleepers - each in the middle of an item - are woken one
//     per unlock, every woken thread waking the next.  (First version of this round: sleepers polled with clock_nanosleep, 50-400 us -
//     the threads that slept through the end of a region cost 1-2 ms per region, 0.62 of the batched rate at 256 threads; round 2, all
//     waiters yielding: 0.47 at 64.)  The cap no longer has to be applied by the caller (GpuSealBfvFactory's callerThreads).
#include <climits>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>
static const int MAX_SPINNERS = 3;
static long futex(std::atomic<int> *addr, int op, int val, const struct timespec *to) { return syscall(SYS_futex, reinterpret_cast<int *>(addr), op, val, to, nullptr, 0); }
void CnMutex::lock(Node &) {
    if (!held.load(std::memory_order_relaxed) && !held.exchange(1, std::memory_order_acquire)) return;
    for (;;) {
        if (spinners.fetch_add(1, std::memory_order_acq_rel) < MAX_SPINNERS) {
            for (int spins = 0;; spins++) {
                if (!held.load(std::memory_order_relaxed) && !held.exchange(1, std::memory_order_acquire)) { spinners.fetch_sub(1, std::memory_order_acq_rel); return; }
                if (spins < 64) __builtin_ia32_pause(); else sched_yield();
            }
        }
        spinners.fetch_sub(1, std::memory_order_acq_rel);
        const int seq = wake_seq.load(std::memory_order_acquire);
        sleepers.fetch_add(1, std::memory_order_acq_rel);
        if (!held.load(std::memory_order_acquire)) {                        // freed meanwhile: do not sleep on a free lock
            sleepers.fetch_sub(1, std::memory_order_acq_rel);
            if (!held.exchange(1, std::memory_order_acquire)) return;
            continue;
        }
        const struct timespec to = {0, 2000000};                            // 2 ms backstop (a wake-up skipped because a spinner existed that then left)
        futex(&wake_seq, FUTEX_WAIT_PRIVATE, seq, &to);
        sleepers.fetch_sub(1, std::memory_order_acq_rel);
        if (!held.load(std::memory_order_relaxed) && !held.exchange(1, std::memory_order_acquire)) return;
    }
}
void CnMutex::unlock(Node &) {
    held.store(0, std::memory_order_release);
    if (sleepers.load(std::memory_order_acquire) > 0 && spinners.load(std::memory_order_acquire) == 0) {
        wake_seq.fetch_add(1, std::memory_order_acq_rel);
        futex(&wake_seq, FUTEX_WAKE_PRIVATE, 1, nullptr);
    }
}

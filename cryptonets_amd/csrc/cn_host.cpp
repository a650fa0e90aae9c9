// Host-only pieces of libcnhip.so shared by every translation unit: the thread-local error message of the C ABI and the context lock.
#include "cn_runtime.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>

static thread_local char g_err[512] = "";
int cn_fail(int code, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return code;
}
extern "C" const char *cn_last_error(void) { return g_err; }
static void err_copy_out(char *dst, size_t n) { snprintf(dst, n, "%s", g_err); }
static void err_set(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); }

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
// Owner bias: a caller thread issues BURSTS of calls (one work item of PoolLayer.Apply is alloc + scalar product + alloc + plain addition + free,
// plus an alloc + encryption per padded tap: 5-15 calls a few hundred nanoseconds apart).  Every critical section works on the same few
// cache lines (queue tails, hazard table, handle table, counters); if the lock changes hands between two calls of a burst those lines
// cross the chip twice per call.  So for `grace` TSC cycles after a release the lock can only be re-taken by the thread that released it;
// everybody else keeps waiting.  The owner's next call finds the lock and the data in its own cache; when it does not come back (the item
// is finished) the others lose the grace period once per item.  A thread that re-acquires more than MAX_BURST times in a row loses the
// privilege for one hand-over (no starvation by a polling loop).  CN_LOCK_GRACE_NS sets the period; DEFAULT 0 = OFF: measured on the
// MI355X box (profiles/r03_unchanged_caller_lock.txt) the bias does not pay - with the literal padded taps the unchanged caller ran at
// 0.85 / 0.85 / 0.84 / 0.82 of the batched rate at 1 / 4 / 16 / 256 threads without it and at 0.84 / 0.84 / 0.78 / 0.72 (400 ns),
// 0.86 / 0.77 / 0.51 / 0.80 (1 us) with it: the threads that lose the grace period queue up behind a holder that is not coming back.
#if defined(__x86_64__) || defined(__i386__)
#include <x86intrin.h>
static inline uint64_t cycle_counter() { return __rdtsc(); }
static inline void cpu_relax() { __builtin_ia32_pause(); }
#else
static inline uint64_t cycle_counter() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; }
static inline void cpu_relax() { sched_yield(); }
#endif
static const uint32_t MAX_BURST = 256;
static std::atomic<uint32_t> g_next_tid{1};
static thread_local uint32_t t_tid = 0;
static uint32_t my_tid() { if (!t_tid) t_tid = g_next_tid.fetch_add(1, std::memory_order_relaxed); return t_tid; }
static uint64_t grace_cycles() {
    static const uint64_t g = [] {
        const char *e = getenv("CN_LOCK_GRACE_NS");
        const double ns = e ? atof(e) : 0.0;
        if (ns <= 0) return (uint64_t)0;
        struct timespec a, b;                                          // TSC cycles per nanosecond, measured over ~2 ms
        clock_gettime(CLOCK_MONOTONIC, &a); const uint64_t c0 = cycle_counter();
        do clock_gettime(CLOCK_MONOTONIC, &b); while ((b.tv_sec - a.tv_sec) * 1000000000ll + (b.tv_nsec - a.tv_nsec) < 2000000);
        const uint64_t c1 = cycle_counter();
        const double per_ns = (double)(c1 - c0) / (double)((b.tv_sec - a.tv_sec) * 1000000000ll + (b.tv_nsec - a.tv_nsec));
        return (uint64_t)(ns * per_ns);
    }();
    return g;
}
bool CnMutex::try_take(uint32_t me) {
    if (held.load(std::memory_order_relaxed)) return false;
    const uint64_t g = grace_cycles();
    if (g && last_owner.load(std::memory_order_relaxed) != me && burst.load(std::memory_order_relaxed) < MAX_BURST &&
        cycle_counter() - released_at.load(std::memory_order_relaxed) < g) return false;                 // the releasing thread may still come back
    if (held.exchange(1, std::memory_order_acquire)) return false;
    if (last_owner.load(std::memory_order_relaxed) == me) burst.fetch_add(1, std::memory_order_relaxed);
    else { last_owner.store(me, std::memory_order_relaxed); burst.store(0, std::memory_order_relaxed); }
    return true;
}
void CnMutex::lock(Node &) {
    const uint32_t me = my_tid();
    if (try_take(me)) return;
    for (;;) {
        if (spinners.fetch_add(1, std::memory_order_acq_rel) < MAX_SPINNERS) {
            for (int spins = 0;; spins++) {
                if (try_take(me)) { spinners.fetch_sub(1, std::memory_order_acq_rel); return; }
                if (spins < 64) cpu_relax(); else sched_yield();
            }
        }
        spinners.fetch_sub(1, std::memory_order_acq_rel);
        const int seq = wake_seq.load(std::memory_order_acquire);
        sleepers.fetch_add(1, std::memory_order_seq_cst);
        if (!held.load(std::memory_order_seq_cst)) {                        // freed meanwhile: do not sleep on a free lock
            sleepers.fetch_sub(1, std::memory_order_acq_rel);
            if (try_take(me)) return;
            continue;
        }
        const struct timespec to = {0, 2000000};                            // 2 ms backstop (a wake-up skipped because a spinner existed that then left)
        const long wrc = futex(&wake_seq, FUTEX_WAIT_PRIVATE, seq, &to);
        sleepers.fetch_sub(1, std::memory_order_acq_rel);
        if (try_take(me)) return;
        static const bool timeout_spin = !(getenv("CN_LOCK_TIMEOUT_SPIN") && !atoi(getenv("CN_LOCK_TIMEOUT_SPIN")));       // A/B switch (default on)
        if (wrc != 0 && timeout_spin) {                                      // timed out (not woken): join the spinners for one bounded round so that a sleeper makes
            spinners.fetch_add(1, std::memory_order_acq_rel);               // progress even while three hot threads keep the spinner slots busy
            for (int spins = 0; spins < 256; spins++) {
                if (try_take(me)) { spinners.fetch_sub(1, std::memory_order_acq_rel); return; }
                if (spins < 64) cpu_relax(); else sched_yield();
            }
            spinners.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
}
void CnMutex::unlock(Node &) {
    if (grace_cycles()) released_at.store(cycle_counter(), std::memory_order_relaxed);
    // seq_cst exchange, not a release store: the store of `held` must not pass the load of `sleepers` (store-load reordering is what x86 does) -
    // a waiter does sleepers++ then reads held, this thread writes held then reads sleepers; without the full barrier both could see the old value and
    // the wake-up would be lost until the futex timeout
    held.exchange(0, std::memory_order_seq_cst);
    if (sleepers.load(std::memory_order_seq_cst) > 0 && spinners.load(std::memory_order_acquire) == 0) {
        wake_seq.fetch_add(1, std::memory_order_acq_rel);
        futex(&wake_seq, FUTEX_WAKE_PRIVATE, 1, nullptr);
    }
}

// ---- combining (CnMutex::run, cn_runtime.h)
bool CnMutex::try_take_me() { return try_take(my_tid()); }
// Combining is OFF by default (CN_LOCK_COMBINE=1 switches it on): measured on the MI355X box (profiles/r03_unchanged_caller_combining.txt) it LOST -
// the unchanged CryptoNets caller ran at 0.92 / 0.77 / 0.58 / 0.57 / 0.47 of the batched rate at 1 / 4 / 16 / 64 / 256 threads with it
// against 0.95 / 0.92 / 0.94 / 0.91 / 0.92 with the plain lock in the same visit: a thread whose request is executed by somebody else still
// has to notice, and a caller thread issues its 5-15 calls per work item one after the other - every one of them then pays a hand-back
// (a cache line pulled across, or a futex wake-up) where the plain lock lets it run its own short critical section as soon as the lock is free.
bool CnMutex::combining() { static const bool on = getenv("CN_LOCK_COMBINE") && atoi(getenv("CN_LOCK_COMBINE")) != 0; return on; }
void CnMutex::release() { Node n; unlock(n); }
// executes every published request (oldest first), a bounded number of rounds; the caller holds the lock
void CnMutex::serve() {
    char saved[sizeof g_err]; bool have_saved = false;
    for (int round = 0; round < 256; round++) {
        CnReq *list = pending.exchange(nullptr, std::memory_order_acquire);
        if (!list) break;
        if (!have_saved) { memcpy(saved, g_err, sizeof g_err); have_saved = true; }       // this thread's own cn_last_error() survives the service
        CnReq *rev = nullptr;
        while (list) { CnReq *nx = list->next; list->next = rev; rev = list; list = nx; }
        while (rev) {
            CnReq *nx = rev->next;                              // (the request may vanish the moment it is marked done)
            const int rc = rev->fn(rev->arg);
            rev->rc = rc;
            if (rc) err_copy_out(rev->err, sizeof rev->err);     // the message was written on THIS thread: hand it to the caller
            const int asleep = rev->asleep.load(std::memory_order_acquire);
            std::atomic<int> *flag = &rev->done;
            flag->store(1, std::memory_order_release);
            if (asleep) futex(flag, FUTEX_WAKE_PRIVATE, 1, nullptr);
            rev = nx;
        }
    }
    if (have_saved) memcpy(g_err, saved, sizeof g_err);
}
int CnMutex::submit(CnReq &r) {
    r.next = pending.load(std::memory_order_relaxed);
    while (!pending.compare_exchange_weak(r.next, &r, std::memory_order_release, std::memory_order_relaxed)) {}
    const uint32_t me = my_tid();
    for (int spins = 0;; spins++) {
        if (r.done.load(std::memory_order_acquire)) break;
        if (try_take(me)) { serve(); release(); continue; }     // every request taken by an earlier holder was completed before it released: mine is done now
        if (spins < 256) cpu_relax();
        else {                                                   // sleep on the request itself; woken by the thread that serves it (50 us backstop)
            r.asleep.store(1, std::memory_order_release);
            if (!r.done.load(std::memory_order_acquire)) { const struct timespec to = {0, 50000}; futex(&r.done, FUTEX_WAIT_PRIVATE, 0, &to); }
            r.asleep.store(0, std::memory_order_relaxed);
        }
    }
    if (r.rc) err_set(r.err);
    return r.rc;
}

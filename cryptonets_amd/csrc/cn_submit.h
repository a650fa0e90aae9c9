// Lock-free submission of per-ciphertext calls (cn_set_option("defer", 2)): the data structures.  Host code only.
//
// The reference calls the wrapper one ciphertext at a time from Defaults.ThreadCount = Environment.ProcessorCount threads (HE Wrapper/Utils.cs:46-88,
// Defaults.cs:11-15; 256 on the bench box, 16 granted cores).  With "defer" = 1 every such call takes the context lock for a few hundred nanoseconds of
// bookkeeping - ~5 700 acquisitions per plaintext prime and CryptoNets batch, and the lock hand-overs (cache lines crossing the chip, futex sleeps and
// wake-ups once more threads wait than spin) were what the unchanged caller spent its time on: 0.91 / 0.86 / 0.72 of the batched rate at 4 / 16 / 256
// threads on the same box, 4-6 CPU-seconds per second (profiles/r06_unchanged_caller_diagnosis.txt).  With "defer" = 2 a deferrable call does not take the
// lock at all: it claims the next slot of a bounded multi-producer ring with one fetch_add, writes a 64-byte record (handles and indices as the caller
// passed them, nothing resolved) and publishes it.  Whoever finds the context lock FREE afterwards drains the ring - executes the records in claim
// order under the lock, on one core, with the hazard table and the queue tails hot in its cache - while the other callers go on publishing; nobody
// ever waits for the lock on this path.  The claim order is a total order consistent with happens-before (a thread that has seen another thread's call
// return claims a later slot), which is all the dependency tracking of the deferred queue needs.
//   * Arguments are checked when the record is executed, not when the call returns: an error surfaces at the next call that synchronises with the
//     context (cn_sync, downloads, any non-deferrable entry point), like an asynchronous device error.  "defer" = 1 keeps the checked-at-the-call
//     behaviour.
//   * Single-ciphertext allocations (AllocateCiphertext of the wrapper: every result of every call) come out of a ring of READY handles the drainer
//     keeps filled (single producer under the lock, lock-free consumers); releases are records like the calls (a release must not overtake a call that
//     reads the array).
#pragma once
#include "../../include/cnhip.h"
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <sched.h>

enum { SUB_FREE = 1, SUB_FREE_MANY, SUB_SCALAR_DOT, SUB_ADD, SUB_SUB, SUB_ADD_PLAIN, SUB_MUL_RELIN, SUB_ENCRYPT, SUB_ENCRYPT_ZERO };
struct alignas(64) SubRec {
    std::atomic<uint64_t> seq;       // slot i starts at i; == pos: free for the producer that claimed pos; == pos + 1: published; the consumer sets pos + CAP
    uint32_t type, count;
    cn_handle a, b, out;             // operands as the caller named them (b: second ciphertext / plaintext array)
    uint32_t ai, bi, oi, x;          // first indices; x: stride of a / subtract flag / K / number of handles
    uint64_t arg;                    // stride of b / seed / payload block (scalar product: K handles, K indices, K weights; release of many: the handles)
};
static_assert(sizeof(SubRec) == 64, "one cache line per record");

struct SubmitRing {
    static constexpr uint64_t CAP = 1ull << 16;
    alignas(64) std::atomic<uint64_t> tail{0};          // next position to claim (producers)
    alignas(64) std::atomic<uint64_t> head{0};          // next position to execute (written by the consumer under the context lock, read by everybody)
    SubRec *rec = nullptr;
    SubmitRing() {
        void *p = nullptr;
        if (posix_memalign(&p, 64, CAP * sizeof(SubRec)) != 0) abort();
        rec = static_cast<SubRec *>(p);
        for (uint64_t i = 0; i < CAP; i++) { new (&rec[i]) SubRec(); rec[i].seq.store(i, std::memory_order_relaxed); }
    }
    ~SubmitRing() { free(rec); }
    SubmitRing(const SubmitRing &) = delete;
    SubmitRing &operator=(const SubmitRing &) = delete;
    bool empty() const { return head.load(std::memory_order_acquire) == tail.load(std::memory_order_acquire); }
    // producer: claim a position; the slot is writable once the record of the previous lap has been executed (full ring: the caller helps draining)
    uint64_t claim() { return tail.fetch_add(1, std::memory_order_acq_rel); }
    SubRec &slot(uint64_t pos) { return rec[pos & (CAP - 1)]; }
    bool writable(uint64_t pos) { return slot(pos).seq.load(std::memory_order_acquire) == pos; }
    void publish(uint64_t pos) { slot(pos).seq.store(pos + 1, std::memory_order_release); }
    // consumer (context lock held): the record at the head if it has been published
    SubRec *peek() { const uint64_t h = head.load(std::memory_order_relaxed); SubRec &s = slot(h); return s.seq.load(std::memory_order_acquire) == h + 1 ? &s : nullptr; }
    // anybody: is the record at the head published?  (a hint for producers: somebody should drain)
    bool peek_published() { const uint64_t h = head.load(std::memory_order_acquire); return slot(h).seq.load(std::memory_order_acquire) == h + 1; }
    void pop() { const uint64_t h = head.load(std::memory_order_relaxed); slot(h).seq.store(h + CAP, std::memory_order_release); head.store(h + 1, std::memory_order_release); }
};

// handles of single size-2 ciphertext arrays that are allocated (live in the handle table) and belong to nobody yet
struct ReadyRing {
    static constexpr uint64_t CAP = 1ull << 13;
    alignas(64) std::atomic<uint64_t> head{0};          // consumers (any thread, compare-and-swap)
    alignas(64) std::atomic<uint64_t> tail{0};          // producer (the drainer, under the context lock)
    std::atomic<uint32_t> misses{0};                     // pops that found the ring empty since the last refill: the target grows with them
    uint32_t target = 64;
    std::atomic<uint64_t> slots[CAP];
    ReadyRing() { for (uint64_t i = 0; i < CAP; i++) slots[i].store(0, std::memory_order_relaxed); }
    uint64_t size() const { const uint64_t t = tail.load(std::memory_order_acquire), h = head.load(std::memory_order_acquire); return t > h ? t - h : 0; }
    cn_handle pop() {
        uint64_t h = head.load(std::memory_order_acquire);
        for (;;) {
            if (h >= tail.load(std::memory_order_acquire)) { misses.fetch_add(1, std::memory_order_relaxed); return 0; }
            const cn_handle v = slots[h & (CAP - 1)].load(std::memory_order_relaxed);      // (a stale read is discarded: the exchange below fails when head has moved)
            if (head.compare_exchange_weak(h, h + 1, std::memory_order_acq_rel, std::memory_order_acquire)) return v;
        }
    }
    bool push(cn_handle v) {                             // producer only
        const uint64_t t = tail.load(std::memory_order_relaxed);
        if (t - head.load(std::memory_order_acquire) >= CAP) return false;
        slots[t & (CAP - 1)].store(v, std::memory_order_relaxed);
        tail.store(t + 1, std::memory_order_release);
        return true;
    }
};

// Stress test of the lock-free submission structures of libcnhip (cryptonets_amd/csrc/cn_submit.h: SubmitRing, ReadyRing) with the protocol
// cn_api.hip runs on them (ring_push / ring_drain / ring_sync / ready_refill) - host code only, run on the CPU by tests/test_host_lock.py.
//   * N producer threads publish M records each; whoever finds the lock free drains.  Every record is executed exactly once, in claim order, records of one
//     thread in program order; a record published BEFORE a release-acquire hand-over to another thread is executed before that thread's next record.
//   * the ring is far smaller than the number of records in flight: full-ring producers help draining, nobody dead-locks.
//   * ReadyRing: one refiller (under the lock), N consumers; every value is handed out exactly once.
#include "../../cryptonets_amd/csrc/cn_runtime.h"
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

struct Ctx {
    CnMutex mu; SubmitRing ring; ReadyRing ready;
    uint64_t executed = 0, last_pos_thread[512] = {0}; long bad = 0;
    uint64_t next_handle = 1;
    uint64_t max_baton_executed = 0;       // hand-over test: the largest value a record carried in `out` among the executed ones
};
static void refill(Ctx &c) { while (c.ready.size() < 256) if (!c.ready.push(c.next_handle++)) break; }
static void exec(Ctx &c, const SubRec &r) {
    // r.a = thread id, r.arg = per-thread sequence number, r.out = the value this record hands over, r.b = the hand-over value its producer had SEEN before it claimed the slot
    if (r.arg != c.last_pos_thread[r.a] + 1) c.bad++;
    c.last_pos_thread[r.a] = r.arg;
    if (r.b > c.max_baton_executed) c.bad++;              // the record that carried baton value r.b must have been executed already
    if (r.out > c.max_baton_executed) c.max_baton_executed = r.out;
    c.executed++;
}
static void drain(Ctx &c, uint64_t upto) {
    for (;;) {
        SubRec *r = c.ring.peek();
        if (!r) {
            if (upto == ~0ull || c.ring.head.load(std::memory_order_relaxed) >= upto) break;
            while (!(r = c.ring.peek())) sched_yield();
        }
        exec(c, *r); c.ring.pop();
    }
    refill(c);
}
static void push(Ctx &c, uint32_t tid, uint64_t seq, uint64_t seen_baton, uint64_t my_baton) {
    const uint64_t pos = c.ring.claim();
    while (!c.ring.writable(pos)) { if (c.mu.try_lock()) { drain(c, ~0ull); c.mu.unlock_now(); } else sched_yield(); }
    SubRec &r = c.ring.slot(pos);
    r.type = 1; r.count = 1; r.a = tid; r.b = seen_baton; r.out = my_baton; r.arg = seq;
    c.ring.publish(pos);
    while (c.ring.peek_published() && c.mu.try_lock()) { drain(c, ~0ull); c.mu.unlock_now(); }
}
int main() {
    for (int threads : {1, 4, 16, 64, 256}) {
        Ctx *cp = new Ctx(); Ctx &c = *cp;
        const int per = 400000 / threads;
        std::atomic<long> popped{0}; std::vector<std::vector<uint64_t>> got(threads);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; t++) ts.emplace_back([&, t] {
            for (int i = 1; i <= per; i++) {
                push(c, (uint32_t)t, (uint64_t)i, 0, 0);
                if (i % 5 == 0) {                                                                        // allocation: ready ring, locked fallback
                    uint64_t h = c.ready.pop();
                    if (!h) { CnGuard g(c.mu); drain(c, c.ring.tail.load()); h = c.next_handle++; refill(c); }
                    got[t].push_back(h); popped++;
                }
                if (i % 1000 == 0) { CnGuard g(c.mu); drain(c, c.ring.tail.load(std::memory_order_acquire)); }   // a synchronising call
            }
        });
        for (auto &t : ts) t.join();
        { CnGuard g(c.mu); drain(c, c.ring.tail.load()); }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / ((double)per * threads);
        std::vector<uint64_t> all; for (auto &v : got) all.insert(all.end(), v.begin(), v.end());
        std::sort(all.begin(), all.end());
        const bool dup = std::adjacent_find(all.begin(), all.end()) != all.end();
        printf("threads %d executed %llu expect %llu bad %ld handles %zu dup %d us_per_call %.3f\n", threads, (unsigned long long)c.executed,
               (unsigned long long)per * threads, c.bad, all.size(), (int)dup, us);
        if (c.executed != (uint64_t)per * threads || c.bad || dup) return 1;
        delete cp;
    }
    // hand-over order: thread A publishes record k and then releases `flag` = k; thread B acquires flag and publishes - B's record must run behind A's
    {
        Ctx *cp = new Ctx(); Ctx &c = *cp;
        std::atomic<uint64_t> flag{0};
        const uint64_t rounds = 200000;
        std::thread a([&] { for (uint64_t k = 1; k <= rounds; k++) { push(c, 0, k, 0, k); flag.store(k, std::memory_order_release); } });
        std::thread b([&] { for (uint64_t k = 1; k <= rounds; k++) { const uint64_t seen = flag.load(std::memory_order_acquire); push(c, 1, k, seen, 0); } });
        std::thread d([&] { for (int i = 0; i < 2000; i++) { { CnGuard g(c.mu); drain(c, c.ring.tail.load(std::memory_order_acquire)); } std::this_thread::yield(); } });
        a.join(); b.join(); d.join();
        { CnGuard g(c.mu); drain(c, c.ring.tail.load()); }
        printf("handover executed %llu expect %llu bad %ld\n", (unsigned long long)c.executed, (unsigned long long)(2 * rounds), c.bad);
        if (c.executed != 2 * rounds || c.bad) return 1;
        delete cp;
    }
    return 0;
}

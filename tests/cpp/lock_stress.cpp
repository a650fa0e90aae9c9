// Stress test of libcnhip's context lock (cryptonets_amd/csrc/cn_host.cpp: bounded spinners, futex sleepers, combining) - host code only, run on
// the CPU by tests/test_host_lock.py: N threads x M critical sections on a plain counter, through CnGuard, through CnMutex::run (the
// closure may be executed by another thread) and mixed; mutual exclusion, no lost wake-up (the run ends), error hand-back, and the time per
// critical section must not collapse with the thread count (the reference calls from Environment.ProcessorCount threads).
#include "../../cryptonets_amd/csrc/cn_runtime.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
extern "C" const char *cn_last_error(void);
int main() {
    for (int mode = 0; mode < 3; mode++) {                    // 0: CnGuard, 1: run(), 2: both mixed
        double base = 0;
        for (int threads : {1, 4, 16, 64, 256}) {
            CnMutex mu; long counter = 0; const int per = 200000 / threads; std::atomic<int> bad{0};
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> ts;
            for (int t = 0; t < threads; t++) ts.emplace_back([&, t] {
                for (int i = 0; i < per; i++) {
                    if (mode == 0 || (mode == 2 && (i + t) % 3 == 0)) { CnGuard g(mu); counter++; for (volatile int w = 0; w < 50; w++) {} }
                    else {
                        const int want = (i % 97 == 0) ? -7 : 0;   // some calls fail: the message must reach THIS thread
                        const int rc = mu.run([&]() -> int { counter++; for (volatile int w = 0; w < 50; w++) {} return want ? cn_fail(want, "fail %d of thread %d", i, t) : 0; });
                        if (rc != want) bad++;
                        if (want) { char exp[64]; snprintf(exp, sizeof exp, "fail %d of thread %d", i, t); if (strcmp(cn_last_error(), exp)) bad++; }
                    }
                }
            });
            for (auto &t : ts) t.join();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / ((double)per * threads);
            printf("mode %d threads %d counter %ld expect %ld bad %d us_per_section %.3f\n", mode, threads, counter, (long)per * threads, bad.load(), us);
            if (counter != (long)per * threads || bad) return 1;
            if (threads == 1) base = us;
            if (us > 100 * base + 20.0) { printf("COLLAPSE at %d threads\n", threads); return 2; }      // (lost wake-ups would show as the 50 us - 2 ms backstops: orders of magnitude)
        }
    }
    return 0;
}

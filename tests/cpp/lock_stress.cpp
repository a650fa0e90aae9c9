// Stress test of libcnhip's context lock (cryptonets_amd/csrc/cn_host.cpp: bounded spinners, futex sleepers) - host code only, run on the CPU by
// tests/test_host_lock.py: N threads x M critical sections on a plain counter; mutual exclusion, no lost wake-up (the run ends), and the
// time per critical section must not collapse with the thread count (the reference calls from Environment.ProcessorCount threads).
#include "../../cryptonets_amd/csrc/cn_runtime.h"
#include <chrono>
#include <cstdio>
#include <thread>
int main() {
    double base = 0;
    for (int threads : {1, 4, 16, 64, 256}) {
        CnMutex mu; long counter = 0; const int per = 200000 / threads;
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; t++) ts.emplace_back([&] { for (int i = 0; i < per; i++) { CnGuard g(mu); counter++; for (volatile int w = 0; w < 50; w++) {} } });
        for (auto &t : ts) t.join();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / ((double)per * threads);
        printf("threads %d counter %ld expect %ld us_per_section %.3f\n", threads, counter, (long)per * threads, us);
        if (counter != (long)per * threads) return 1;
        if (threads == 1) base = us;
        if (us > 20 * base + 1.0) { printf("COLLAPSE at %d threads\n", threads); return 2; }
    }
    return 0;
}

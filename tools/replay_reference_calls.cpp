// The reference's UNCHANGED caller, restated in C++ on top of the C ABI (include/cnhip.h): the literal per-ciphertext call pattern the C#
// NeuralNetworks layers issue through the HE-wrapper twin (integration/*.cs), from Defaults.ThreadCount threads.
//
//   PoolLayer.Apply (NeuralNetworks/PoolLayer.cs:149-229): ParallelProcessInEnv(maps * corners, k => {
//        conv   = ConvolveOnce(m, corner, map)  -> patch.Mul(weightWindows[map])                       (:113-121)
//                 -> EncryptedSealBfvVector.DenseMatrixBySparseVectorMultiply -> one Task per plaintext prime (EncryptedSealBfvVector.cs:225-236)
//                 -> AtomicSealBfvEncryptedVector.DenseMatrixBySparseVectorMultiply (AtomicSealBfvVector.cs:434-521)  = cn_scalar_dot
//        res[k] = conv.Add(biasVectors[map])    -> Evaluator.AddPlain (AtomicSealBfvVector.cs:1019)                     = cn_add_plain
//        conv.Dispose()                                                                                                  = cn_free })
//   SquareActivation.Apply (SquareActivation.cs:10-13) -> EncryptedSealBfvMatrix.ElementWiseMultiply (EncryptedSealBfvMatrix.cs:140-154):
//        ParallelProcessInEnv(columns, k => PointwiseMultiply -> Multiply + Relinearize (AtomicSealBfvVector.cs:839-840)  = cn_mul_relin, count 1)
//   BaseLayer.GetNext (BaseLayer.cs:23-49) disposes a layer's input matrix once its output exists                         = cn_free per column
//   PoolLayer.ElementAt (PoolLayer.cs:67-80), `literal_taps`: a convolution tap that falls into the padding is a FRESH encryption of the
//        zero vector - Factory.GetEncryptedVector(zeros) per (map, corner, padded offset), kept in TempVectors and disposed by
//        ReleaseTemp() after the layer (:83-90, :189,226)                                              = cn_ct_alloc + cn_encrypt(pt = 0) ... cn_free
//        (645 per plaintext prime and batch for the 5x5 stride-2 convolution of CryptoNets; without `literal_taps` the tap is passed as
//        handle 0 = skipped - the batched path's deviation, identical plaintext)
//   Utils.ParallelProcessInEnv (HE Wrapper/Utils.cs:46-88): `threads` tasks pulling item indices from an interlocked counter; the tasks run
//        on pooled threads (.NET thread pool) - here a pool of `threads` workers that lives as long as the library.
//
// Every ciphertext is its own handle (count 1), exactly like the individually allocated SEAL Ciphertext objects of the reference.
// Used by tools/replay_reference_calls.py, bench.py (`unchanged_caller`) and tests/test_deferred.py; measurement tooling, not product.
//   g++ -O2 -std=c++17 -shared -fPIC tools/replay_reference_calls.cpp -Iinclude -Lcryptonets_amd/lib -lcnhip -Wl,-rpath,'$ORIGIN' -pthread
#include "../include/cnhip.h"
#include <atomic>
#include <climits>
#include <cstdio>
#include <linux/futex.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <algorithm>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" {
typedef struct rp_layer {
    uint32_t O, K;             // outputs, taps per output
    const int32_t *idx;        // [O][K] input column of every tap, -1 = padded tap (ConvolutionEngine: outside the image)
    const uint64_t *W;         // [primes][O][K] weight residues mod t_p (sparse plain weight windows, PoolLayer.cs:101-111)
    const cn_handle *bias_pt;  // [primes] dense plaintext array holding the bias vectors of the layer (biasVectors, :158-171), or NULL
    const int32_t *bias_idx;   // [O] plaintext of output o inside bias_pt
    int square;                // a SquareActivation follows
} rp_layer;
}

namespace {
struct Err { std::atomic<int> rc{0}; char msg[512] = ""; };
void note(Err &e, int rc) {
    int expect = 0;
    if (rc && e.rc.compare_exchange_strong(expect, rc)) snprintf(e.msg, sizeof e.msg, "%s", cn_last_error());
}
// Utils.ParallelProcessInEnv: `threads` tasks (Task.Run on the .NET thread pool), work items handed out by an interlocked counter.  The
// pool threads are created once and reused by every region, like the runtime's pool.
class Pool {
    std::vector<std::thread> workers;
    const std::function<void(int)> *body = nullptr;
    // ONE word tells a pool thread everything about a region: (generation << 12) | participants.  (Rounds 3-6 kept the participant count in a second
    // atomic: a thread that did not take part in region G could read G's generation, lose the CPU, read the participant count of region G + 1, work in
    // G + 1 - and then see G + 1's generation as new and take part in it a SECOND time: `running` reached zero one thread early and run() returned while
    // a body was still executing.  Found by tools/soak_lockfree.py, which changes the thread count from run to run; one read of one word cannot tear.)
    static constexpr int PART_BITS = 12, PART_MASK = (1 << PART_BITS) - 1;
    std::atomic<int> word{0}, next{0}, running{0}, done_flag{0};
    int count = 0, generation = 0; std::atomic<bool> stop{false};
    static long fut(std::atomic<int> *a, int op, int v) { return syscall(SYS_futex, reinterpret_cast<int *>(a), op, v, nullptr, nullptr, 0); }
    void work(int id) {
        int seen = 0;
        for (;;) {
            for (int spins = 0;; spins++) {                                   // idle pool thread: spin briefly, then sleep on the region word
                const int w = word.load(std::memory_order_acquire);
                if (stop.load()) return;
                if (w != seen) { seen = w; if (id < (w & PART_MASK)) break; else continue; }
                if (spins < 200) __builtin_ia32_pause(); else fut(&word, FUTEX_WAIT_PRIVATE, w);
            }
            for (;;) { const int k = next.fetch_add(1); if (k >= count) break; (*body)(k); }
            if (running.fetch_sub(1, std::memory_order_acq_rel) == 1) { done_flag.store(1, std::memory_order_release); fut(&done_flag, FUTEX_WAKE_PRIVATE, 1); }
        }
    }
public:
    ~Pool() { stop = true; word.fetch_add(1 << PART_BITS); fut(&word, FUTEX_WAKE_PRIVATE, INT_MAX); for (auto &t : workers) t.join(); }
    void run(int n_items, int threads, const std::function<void(int)> &fn) {
        if (n_items < 2 || threads < 2) { for (int k = 0; k < n_items; k++) fn(k); return; }
        const int nt = std::min(threads > n_items ? n_items : threads, (int)PART_MASK);
        while ((int)workers.size() < nt) { const int id = (int)workers.size(); workers.emplace_back([this, id] { work(id); }); }
        body = &fn; count = n_items; next.store(0); running.store(nt); done_flag.store(0);
        generation = (generation + 1) & 0x7FFFF;
        word.store((generation << PART_BITS) | nt, std::memory_order_release);
        fut(&word, FUTEX_WAKE_PRIVATE, INT_MAX);
        while (!done_flag.load(std::memory_order_acquire)) fut(&done_flag, FUTEX_WAIT_PRIVATE, 0);
    }
};
// What the twin does INSIDE the classes it owns (integration/GpuAtomicSealBfvEncryptedVector.cs), mode bit 1 of `literal_taps`:
//   * a zero vector is made with ONE library call (cn_encrypt_zero_new = AllocateCiphertext + Encrypt(PlainZero)),
//   * Dispose() of a device array parks the handle in a per-thread list that is released with one cn_free_many per 32 handles
//     (CnBuffer.Free -> CnDevice.DeferFree); what is left is released at the end of the inference (Download / Decrypt flush every list).
//     Exception: the result of DenseMatrixBySparseVectorMultiply (CnBuffer.FreeNow) is released at once - the queue folds `conv.Add(bias)` into the
//     scalar product only for an intermediate it knows to be dead (parked, 5-25 of the 100 outputs of the dense layer missed the fold and the layer
//     was cut into two GEMM launches, profiles/HISTORY.md round 4).
// The unchanged layers above the twin make exactly the same calls as before.
struct FreeBin { cn_ctx *ctx; std::vector<cn_handle> h; };
struct FreeBins {
    std::mutex mu; std::vector<std::vector<FreeBin> *> all;
    void add(std::vector<FreeBin> *b) { std::lock_guard<std::mutex> g(mu); all.push_back(b); }
    int flush_all() {
        std::lock_guard<std::mutex> g(mu);
        int rc = 0;
        for (auto *bins : all) for (FreeBin &b : *bins) if (!b.h.empty()) { const int r = cn_free_many(b.ctx, b.h.data(), (uint32_t)b.h.size()); if (r && !rc) rc = r; b.h.clear(); }
        return rc;
    }
};
FreeBins &free_bins() { static FreeBins f; return f; }
int defer_free(cn_ctx *ctx, cn_handle h) {
    thread_local std::vector<FreeBin> *bins = nullptr;
    if (!bins) { bins = new std::vector<FreeBin>(); free_bins().add(bins); }
    FreeBin *b = nullptr;
    for (FreeBin &x : *bins) if (x.ctx == ctx) b = &x;
    if (!b) { bins->push_back(FreeBin{ctx, {}}); b = &bins->back(); }
    b->h.push_back(h);
    if (b->h.size() < 32) return 0;
    const int rc = cn_free_many(ctx, b->h.data(), (uint32_t)b->h.size());
    b->h.clear();
    return rc;
}
Pool &pool() { static Pool p; return p; }
void parallel_process(int count, int threads, const std::function<void(int)> &body) { pool().run(count, threads, body); }
}  // namespace

// in:  [primes][n_in] one handle per input column (count 1 each); out: [primes][O_last] receives the handles of the last layer's
// columns (owned by the caller from then on).  The inputs are left alive (the reference's EncryptLayer output would be disposed).
// literal_taps: padded taps are fresh encryptions of zero (needs the public key in the contexts); nonce0: first encryption nonce
extern "C" int rp_run2(cn_ctx **ctx, int nprimes, const rp_layer *layers, int nlayers, const cn_handle *in, uint32_t n_in, cn_handle *out, int threads,
                       int literal_taps, uint64_t nonce0, char *errmsg, size_t errlen) {
    Err err;
    const bool merged = (literal_taps & 2) != 0;                                     // the twin's one-call zero vectors and batched disposal (see FreeBins)
    const bool direct_free = (literal_taps & 4) != 0;                                // ... on a lock-free context ("defer" = 2) Dispose() is one published record: nothing is parked, the
    literal_taps &= 1;                                                               // library sees every release where the caller made it (and can tell which zero vectors are dead)
    auto release = [&](cn_ctx *c, cn_handle h) { return merged && !direct_free ? defer_free(c, h) : cn_free(c, h); };
    std::atomic<uint64_t> nonce{nonce0};
    std::vector<std::vector<cn_handle>> cur(nprimes);
    for (int p = 0; p < nprimes; p++) cur[p].assign(in + (size_t)p * n_in, in + (size_t)(p + 1) * n_in);
    bool cur_owned = false;
    auto dispose = [&](std::vector<std::vector<cn_handle>> &m) {                  // IMatrix.Dispose: every column, every prime
        for (int p = 0; p < nprimes; p++) for (cn_handle h : m[p]) note(err, release(ctx[p], h));
    };
    for (int li = 0; li < nlayers && !err.rc; li++) {
        const rp_layer &L = layers[li];
        std::vector<std::vector<cn_handle>> res(nprimes, std::vector<cn_handle>(L.O, 0));
        std::vector<std::vector<std::pair<int, cn_handle>>> temps(L.O);              // TempVectors: the zero encryptions of item k, per prime
        // ---- PoolLayer.Apply
        parallel_process((int)L.O, threads, [&](int k) {
            if (err.rc) return;
            std::vector<cn_handle> patch(L.K);
            for (int p = 0; p < nprimes; p++) {                                      // ForEveryEncryptedVector: one task per plaintext prime
                int rc = 0;
                for (uint32_t t = 0; t < L.K && !rc; t++) {
                    const int32_t c = L.idx[(size_t)k * L.K + t];
                    if (c >= 0) { patch[t] = cur[p][c]; continue; }
                    patch[t] = 0;
                    if (!literal_taps) continue;
                    cn_handle z = 0;                                                 // ElementAt: Factory.GetEncryptedVector(zeros, dense, m.Scale)
                    if (merged) rc = cn_encrypt_zero_new(ctx[p], nonce.fetch_add(1), &z);
                    else {
                        rc = cn_ct_alloc(ctx[p], 1, 2, &z);
                        if (!rc) rc = cn_encrypt(ctx[p], 0, 0, 0, z, 0, 1, nonce.fetch_add(1));
                    }
                    if (!rc) { patch[t] = z; temps[k].emplace_back(p, z); }
                }
                if (rc) { note(err, rc); return; }
                cn_handle conv = 0, r = 0;
                rc = cn_ct_alloc(ctx[p], 1, 2, &conv);                             // AllocateCiphertext(env)
                if (!rc) rc = cn_scalar_dot(ctx[p], patch.data(), nullptr, L.W + ((size_t)p * L.O + k) * L.K, L.K, conv, 0);
                if (!rc && L.bias_pt) {
                    rc = cn_ct_alloc(ctx[p], 1, 2, &r);
                    if (!rc) rc = cn_add_plain(ctx[p], conv, 0, L.bias_pt[p], (uint32_t)L.bias_idx[k], 0, r, 0, 1);
                    if (!rc) rc = cn_free(ctx[p], conv);                           // using (conv) { ... }: released AT ONCE, never parked - the library folds the bias
                                                                                       // addition into the scalar product only when it knows the intermediate is dead
                } else r = conv;
                res[p][k] = r;
                if (rc) { note(err, rc); return; }
            }
        });
        if (literal_taps) {                                                         // ReleaseTemp(): Parallel.ForEach(TempVectors, v => v.Dispose())
            std::vector<std::pair<int, cn_handle>> all;
            for (auto &v : temps) all.insert(all.end(), v.begin(), v.end());
            parallel_process((int)all.size(), threads, [&](int i) { note(err, release(ctx[all[i].first], all[i].second)); });
        }
        if (cur_owned) dispose(cur);                                                // BaseLayer.GetNext: m.Dispose()
        cur.swap(res); cur_owned = true;
        if (err.rc || !L.square) continue;
        // ---- SquareActivation.Apply -> ElementWiseMultiply(m, m)
        std::vector<std::vector<cn_handle>> sq(nprimes, std::vector<cn_handle>(L.O, 0));
        parallel_process((int)L.O, threads, [&](int k) {
            if (err.rc) return;
            for (int p = 0; p < nprimes; p++) {
                cn_handle r = 0;
                int rc = cn_ct_alloc(ctx[p], 1, 2, &r);
                if (!rc) rc = cn_mul_relin(ctx[p], cur[p][k], 0, 1, cur[p][k], 0, 1, r, 0, 1);        // Multiply + Relinearize
                sq[p][k] = r;
                if (rc) { note(err, rc); return; }
            }
        });
        dispose(cur);
        cur.swap(sq);
    }
    if (err.rc) {
        if (cur_owned) dispose(cur);
        (void)free_bins().flush_all();
        if (errmsg && errlen) snprintf(errmsg, errlen, "%s", err.msg);
        return err.rc;
    }
    if (merged) note(err, free_bins().flush_all());                                  // end of the inference: the parked handles of every thread
    const uint32_t O = layers[nlayers - 1].O;
    for (int p = 0; p < nprimes; p++) memcpy(out + (size_t)p * O, cur[p].data(), (size_t)O * sizeof(cn_handle));
    return 0;
}
// Host-only self-test of the pool (no device call): `regions` parallel regions of pseudo-random width and item count.  Returns the number of
// violations: an item executed twice or not at all, a body still running when its region has returned.
extern "C" int rp_pool_selftest(int regions, unsigned seed) {
    std::vector<std::atomic<int>> hits(1024);
    std::atomic<int> alive{0};
    int bad = 0;
    uint64_t x = (uint64_t)seed * 0x9E3779B97F4A7C15ull + 1;
    for (int r = 0; r < regions; r++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const int threads = (r & 1) ? 2 + (int)(x % 199) : 2 + (int)(x % 3), items = (r & 1) ? 2 + (int)((x >> 20) % 400) : 2 + (int)((x >> 20) % 4);
        for (int i = 0; i < items; i++) hits[i].store(0);
        parallel_process(items, threads, [&](int k) {                 // a body that lasts: a region that returns early is caught with a body alive
            alive.fetch_add(1); hits[k].fetch_add(1);
            for (volatile int spin = 0; spin < 2000; spin = spin + 1) {}
            if ((k & 15) == 0) sched_yield();
            alive.fetch_sub(1);
        });
        if (alive.load()) bad++;
        for (int i = 0; i < items; i++) if (hits[i].load() != 1) bad++;
    }
    return bad;
}
extern "C" int rp_run(cn_ctx **ctx, int nprimes, const rp_layer *layers, int nlayers, const cn_handle *in, uint32_t n_in, cn_handle *out, int threads,
                      char *errmsg, size_t errlen) {
    return rp_run2(ctx, nprimes, layers, nlayers, in, n_in, out, threads, 0, 0, errmsg, errlen);
}

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
//   Utils.ParallelProcessInEnv (HE Wrapper/Utils.cs:46-88): `threads` workers pulling item indices from an interlocked counter.
//
// Every ciphertext is its own handle (count 1), exactly like the individually allocated SEAL Ciphertext objects of the reference.
// Used by tools/replay_reference_calls.py, bench.py (`unchanged_caller`) and tests/test_deferred.py; measurement tooling, not product.
//   g++ -O2 -std=c++17 -shared -fPIC tools/replay_reference_calls.cpp -Iinclude -Lcryptonets_amd/lib -lcnhip -Wl,-rpath,'$ORIGIN' -pthread
#include "../include/cnhip.h"
#include <atomic>
#include <cstdio>
#include <cstring>
#include <functional>
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
// Utils.ParallelProcessInEnv: up to `threads` workers, work items handed out by an interlocked counter
void parallel_process(int count, int threads, const std::function<void(int)> &body) {
    if (count < 2 || threads < 2) { for (int k = 0; k < count; k++) body(k); return; }
    std::atomic<int> next{-1};
    const int nt = threads > count ? count : threads;
    std::vector<std::thread> pool;
    pool.reserve(nt);
    for (int t = 0; t < nt; t++) pool.emplace_back([&] { for (;;) { const int k = ++next; if (k >= count) break; body(k); } });
    for (auto &t : pool) t.join();
}
}  // namespace

// in:  [primes][n_in] one handle per input column (count 1 each); out: [primes][O_last] receives the handles of the last layer's
// columns (owned by the caller from then on).  The inputs are left alive (the reference's EncryptLayer output would be disposed).
extern "C" int rp_run(cn_ctx **ctx, int nprimes, const rp_layer *layers, int nlayers, const cn_handle *in, uint32_t n_in, cn_handle *out, int threads,
                      char *errmsg, size_t errlen) {
    Err err;
    std::vector<std::vector<cn_handle>> cur(nprimes);
    for (int p = 0; p < nprimes; p++) cur[p].assign(in + (size_t)p * n_in, in + (size_t)(p + 1) * n_in);
    bool cur_owned = false;
    auto dispose = [&](std::vector<std::vector<cn_handle>> &m) {                  // IMatrix.Dispose: every column, every prime
        for (int p = 0; p < nprimes; p++) for (cn_handle h : m[p]) note(err, cn_free(ctx[p], h));
    };
    for (int li = 0; li < nlayers && !err.rc; li++) {
        const rp_layer &L = layers[li];
        std::vector<std::vector<cn_handle>> res(nprimes, std::vector<cn_handle>(L.O, 0));
        // ---- PoolLayer.Apply
        parallel_process((int)L.O, threads, [&](int k) {
            if (err.rc) return;
            std::vector<cn_handle> patch(L.K);
            for (int p = 0; p < nprimes; p++) {                                      // ForEveryEncryptedVector: one task per plaintext prime
                for (uint32_t t = 0; t < L.K; t++) { const int32_t c = L.idx[(size_t)k * L.K + t]; patch[t] = c < 0 ? 0 : cur[p][c]; }
                cn_handle conv = 0, r = 0;
                int rc = cn_ct_alloc(ctx[p], 1, 2, &conv);                         // AllocateCiphertext(env)
                if (!rc) rc = cn_scalar_dot(ctx[p], patch.data(), nullptr, L.W + ((size_t)p * L.O + k) * L.K, L.K, conv, 0);
                if (!rc && L.bias_pt) {
                    rc = cn_ct_alloc(ctx[p], 1, 2, &r);
                    if (!rc) rc = cn_add_plain(ctx[p], conv, 0, L.bias_pt[p], (uint32_t)L.bias_idx[k], 0, r, 0, 1);
                    if (!rc) rc = cn_free(ctx[p], conv);                           // using (conv) { ... }
                } else r = conv;
                res[p][k] = r;
                if (rc) { note(err, rc); return; }
            }
        });
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
        if (errmsg && errlen) snprintf(errmsg, errlen, "%s", err.msg);
        return err.rc;
    }
    const uint32_t O = layers[nlayers - 1].O;
    for (int p = 0; p < nprimes; p++) memcpy(out + (size_t)p * O, cur[p].data(), (size_t)O * sizeof(cn_handle));
    return 0;
}

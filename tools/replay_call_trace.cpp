// Replays a recorded sequence of libcnhip calls (tools/call_trace.py) from native code, on the C ABI of include/cnhip.h - the host side a
// compiled caller (the C# twin through P/Invoke) presents to the library, without an interpreter between the calls.  Used for the
// UNCHANGED per-call pattern of the LoLa networks (LowLatencyCryptoNets/LoLaCryptonets.cs:203-278 through the reference's unchanged
// EncryptedSealBfvMatrix / LL*Layer files: one AtomicSealBfvEncryptedVector method per row, column and map): bench.py --workload lola
// reports it as `unchanged_caller`.  Measurement tooling, not product.
//
// Trace format (u64 words): [n_records, n_ext, n_new] then per record [opcode, n_ints, blob_words, ints..., blob...].  A handle argument is
// all ones (no handle), bit 63 | i (the i-th handle allocated inside the trace) or i (external handle i, supplied by the driver).  Integer
// arguments follow the positional order of cryptonets_amd._native.Context's methods (defaults filled in).
//   g++ -O2 -std=c++17 -shared -fPIC tools/replay_call_trace.cpp -Iinclude -Lcryptonets_amd/lib -lcnhip -Wl,-rpath,'$ORIGIN' -pthread
#include "../include/cnhip.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace {
enum Op { CT_ALLOC, PT_ALLOC, FREE, COPY, ADD, SUB, NEGATE, ADD_MANY, ADD_PLAIN, MUL_PLAIN, MUL_SCALAR, SCALAR_DOT, MUL_RELIN, ROTATE_ROWS, ROTATE_COLUMNS,
          ROTATE_ROWS_ADD, ROTATE_COLUMNS_ADD, SUM_SLOTS, ENCODE_BATCH, PT_UPLOAD, GEMM_APPLY, MULTIPLY, RELINEARIZE, APPLY_GALOIS, ROWDOT_BATCH, SCALAR_GEMM, COPY_MANY, ROTATE_ROWS_MANY };
struct Rec { uint64_t op; const uint64_t *ints; uint64_t n_ints; const uint64_t *blob; uint64_t blob_words; };
struct Trace {
    cn_ctx *ctx; std::vector<Rec> recs; const uint64_t *ext; uint64_t n_ext, n_new;
    std::vector<cn_handle> made;              // handles allocated by the current repetition (0 = not yet / freed)
    uint64_t next_new = 0;
    cn_handle H(uint64_t w) const { return w == ~0ull ? 0 : ((w >> 63) ? made[w & 0x7fffffffffffffffull] : ext[w]); }
};
const uint64_t NEW_BIT = 1ull << 63;
int run_one(Trace &t, const Rec &r) {
    const uint64_t *a = r.ints;
    auto I = [&](int i) { return (int64_t)a[i]; };
    auto U = [&](int i) { return (uint32_t)a[i]; };
    switch ((Op)r.op) {
    case CT_ALLOC: { cn_handle h = 0; int rc = cn_ct_alloc(t.ctx, U(0), U(1), &h); t.made[t.next_new++] = h; return rc; }
    case PT_ALLOC: { cn_handle h = 0; int rc = cn_pt_alloc(t.ctx, U(0), &h); t.made[t.next_new++] = h; return rc; }
    case FREE: { const uint64_t id = a[0] & ~NEW_BIT; int rc = cn_free(t.ctx, t.made[id]); t.made[id] = 0; return rc; }
    case COPY: return cn_copy(t.ctx, t.H(a[0]), U(1), t.H(a[2]), U(3), U(4));
    case ADD: return cn_add(t.ctx, t.H(a[0]), U(1), t.H(a[2]), U(3), t.H(a[4]), U(5), U(6));
    case SUB: return cn_sub(t.ctx, t.H(a[0]), U(1), t.H(a[2]), U(3), t.H(a[4]), U(5), U(6));
    case NEGATE: return cn_negate(t.ctx, t.H(a[0]), U(1), t.H(a[2]), U(3), U(4));
    case ADD_MANY: { std::vector<uint32_t> idx(r.blob_words); for (uint64_t i = 0; i < r.blob_words; i++) idx[i] = (uint32_t)r.blob[i];
                     return cn_add_many(t.ctx, t.H(a[0]), idx.data(), (uint32_t)idx.size(), t.H(a[1]), U(2)); }
    case ADD_PLAIN: return cn_add_plain(t.ctx, t.H(a[0]), U(1), t.H(a[2]), U(3), (int)I(7), t.H(a[4]), U(5), U(6));       // a ai pt pi out oi count subtract
    case MUL_PLAIN: return cn_mul_plain(t.ctx, t.H(a[0]), U(1), t.H(a[2]), U(3), U(7), t.H(a[4]), U(5), U(6));            // a ai pt pi out oi count pt_stride
    case MUL_SCALAR: return cn_mul_scalar(t.ctx, t.H(a[0]), U(1), r.blob, U(2), t.H(a[3]), U(4), U(5));
    case SCALAR_DOT: { const uint32_t K = U(0); std::vector<cn_handle> hs(K); std::vector<uint32_t> ix(K);
                       for (uint32_t i = 0; i < K; i++) { hs[i] = t.H(r.blob[i]); ix[i] = (uint32_t)r.blob[K + i]; }
                       return cn_scalar_dot(t.ctx, hs.data(), ix.data(), r.blob + 2 * K, K, t.H(a[1]), U(2)); }
    case MUL_RELIN: return cn_mul_relin(t.ctx, t.H(a[0]), U(1), U(7), t.H(a[2]), U(3), U(8), t.H(a[4]), U(5), U(6));      // a ai b bi out oi count a_stride b_stride
    case ROTATE_ROWS: return cn_rotate_rows(t.ctx, t.H(a[0]), U(1), (int)I(2), t.H(a[3]), U(4), U(5));
    case ROTATE_COLUMNS: return cn_rotate_columns(t.ctx, t.H(a[0]), U(1), t.H(a[2]), U(3), U(4));
    case ROTATE_ROWS_ADD: return cn_rotate_rows_add(t.ctx, t.H(a[0]), U(1), (int)I(2), t.H(a[3]), U(4), t.H(a[5]), U(6), U(7));
    case ROTATE_COLUMNS_ADD: return cn_rotate_columns_add(t.ctx, t.H(a[0]), U(1), t.H(a[2]), U(3), t.H(a[4]), U(5), U(6));
    case SUM_SLOTS: return cn_sum_slots(t.ctx, t.H(a[0]), U(1), U(2), U(3));
    case ENCODE_BATCH: return cn_encode_batch(t.ctx, r.blob, U(2), U(3), t.H(a[0]), U(1));                               // pt pi nvalues count
    case PT_UPLOAD: return cn_pt_upload(t.ctx, t.H(a[0]), U(1), U(2), r.blob);
    case GEMM_APPLY: return cn_gemm_plan_apply(t.ctx, t.H(a[0]), t.H(a[1]), t.H(a[2]), U(3));
    case MULTIPLY: return cn_multiply(t.ctx, t.H(a[0]), U(1), t.H(a[2]), U(3), t.H(a[4]), U(5), U(6));
    case RELINEARIZE: return cn_relinearize(t.ctx, t.H(a[0]), U(1), t.H(a[2]), U(3), U(4));
    case APPLY_GALOIS: return cn_apply_galois(t.ctx, t.H(a[0]), U(1), a[2], t.H(a[3]), U(4), U(5));
    case ROWDOT_BATCH: return cn_rowdot_batch(t.ctx, t.H(a[0]), U(1), t.H(a[2]), U(3), U(4), U(5), t.H(a[6]), U(7));
    case COPY_MANY: { const uint32_t n = U(2); std::vector<cn_handle> hs(n); std::vector<uint32_t> fs(n);
                      for (uint32_t i = 0; i < n; i++) { hs[i] = t.H(r.blob[i]); fs[i] = (uint32_t)r.blob[n + i]; }
                      return cn_copy_many(t.ctx, hs.data(), fs.data(), n, t.H(a[0]), U(1)); }
    case ROTATE_ROWS_MANY: { const uint32_t n = U(2); std::vector<uint32_t> ii(n), oi(n); std::vector<int> st(n);
                             for (uint32_t i = 0; i < n; i++) { ii[i] = (uint32_t)r.blob[i]; st[i] = (int)(int64_t)r.blob[n + i]; oi[i] = (uint32_t)r.blob[2 * n + i]; }
                             return cn_rotate_rows_many(t.ctx, t.H(a[0]), ii.data(), st.data(), n, t.H(a[1]), oi.data()); }
    case SCALAR_GEMM: { const uint32_t O = U(1), K = U(2); const int32_t *idx = nullptr; std::vector<int32_t> ix((size_t)O * K), bi(O);
                        for (size_t i = 0; i < (size_t)O * K; i++) ix[i] = (int32_t)(int64_t)r.blob[i];
                        for (uint32_t o = 0; o < O; o++) bi[o] = (int32_t)(int64_t)r.blob[(size_t)2 * O * K + o];
                        idx = ix.data();
                        return cn_scalar_gemm(t.ctx, t.H(a[0]), idx, r.blob + (size_t)O * K, O, K, t.H(a[3]), t.H(a[3]) ? bi.data() : nullptr, t.H(a[4]), U(5)); }
    }
    return CN_ERR_ARG;
}
// release what a repetition allocated and did not free itself (cached temporaries of the recorded program), except `keep`
int cleanup(Trace &t, uint64_t keep, bool keep_it) {
    int rc = 0;
    for (uint64_t i = 0; i < t.made.size(); i++) if (t.made[i] && !(keep_it && i == keep)) { int r2 = cn_free(t.ctx, t.made[i]); if (!rc) rc = r2; t.made[i] = 0; }
    return rc;
}
struct SpinBarrier {
    std::atomic<int> count{0}, gen{0}; int n;
    explicit SpinBarrier(int n_) : n(n_) {}
    void wait() { const int g = gen.load(); if (count.fetch_add(1) + 1 == n) { count.store(0); gen.fetch_add(1); } else while (gen.load() == g) __builtin_ia32_pause(); }
};
}  // namespace

// results[c]: in = trace-local id of the result handle of context c, out = the live handle after the last repetition (caller frees)
extern "C" int ct_replay(cn_ctx **ctx, int n, const uint64_t *const *traces, const uint64_t *const *exts, int warmup, int reps, int mode, double *ms_per_rep,
                         uint64_t *results, char *errmsg, size_t errlen) {
    std::vector<Trace> T(n);
    for (int c = 0; c < n; c++) {
        const uint64_t *w = traces[c];
        T[c].ctx = ctx[c]; T[c].ext = exts[c]; T[c].n_ext = w[1]; T[c].n_new = w[2];
        T[c].made.assign(w[2], 0);
        const uint64_t nrec = w[0]; w += 3;
        for (uint64_t i = 0; i < nrec; i++) { Rec r{w[0], w + 3, w[1], w + 3 + w[1], w[2]}; T[c].recs.push_back(r); w += 3 + w[1] + w[2]; }
    }
    std::atomic<int> err{0}; char msg[512] = "";
    auto fail = [&](int rc, int c, uint64_t i) { int e = 0; if (rc && err.compare_exchange_strong(e, rc)) snprintf(msg, sizeof msg, "context %d record %llu op %llu: %s", c, (unsigned long long)i, (unsigned long long)T[c].recs[i].op, cn_last_error()); };
    const uint64_t nrec = T[0].recs.size();
    for (int c = 1; c < n; c++) if (mode != 2 && T[c].recs.size() != nrec) { snprintf(errmsg, errlen, "the contexts' traces have different lengths (%zu vs %llu): use mode 2", T[c].recs.size(), (unsigned long long)nrec); return CN_ERR_ARG; }
    std::chrono::steady_clock::time_point t0;
    for (int rep = 0; rep < warmup + reps && !err; rep++) {
        if (rep == warmup) { for (int c = 0; c < n; c++) cn_sync(ctx[c]); t0 = std::chrono::steady_clock::now(); }
        for (int c = 0; c < n; c++) T[c].next_new = 0;
        if (mode == 0) {                                   // one host thread, the contexts call by call
            for (uint64_t i = 0; i < nrec && !err; i++) for (int c = 0; c < n; c++) fail(run_one(T[c], T[c].recs[i]), c, i);
        } else {                                           // one thread per context: joined after every call (1) or free running (2)
            SpinBarrier bar(n);
            std::vector<std::thread> th;
            for (int c = 0; c < n; c++) th.emplace_back([&, c] {
                for (uint64_t i = 0; i < T[c].recs.size(); i++) { if (!err) fail(run_one(T[c], T[c].recs[i]), c, i); if (mode == 1) bar.wait(); }
                // the caller reads the result of every inference: Decrypt follows, and the reference runs it - like every operation of the CRT layer - as one task per plaintext
                // prime (EncryptedSealBfvVector.cs:225-236).  Round 6: the wait belongs to the prime's own thread.  (It used to be a loop over the contexts behind the join: with
                // deferred submission the LAST queue segment of a prime is launched by its cn_sync, so the four final segments ran one after the other - 2-3 ms per image
                // of an artefact of this harness, profiles/r06_lola_unchanged_queue_gaps.txt.)
                if (!err) fail(cn_sync(ctx[c]), c, 0);
            });
            for (auto &x : th) x.join();
        }
        if (mode == 0) {                                   // one host thread: launch what every context has queued (cn_stats_get flushes without waiting), THEN wait
            cn_stats st;
            for (int c = 0; c < n; c++) cn_stats_get(ctx[c], &st, 0);
        }
        for (int c = 0; c < n; c++) cn_sync(ctx[c]);
        const bool last = rep == warmup + reps - 1;
        for (int c = 0; c < n; c++) { int rc = cleanup(T[c], results[c], last); if (rc) fail(rc, c, 0); }
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (err) { if (errmsg && errlen) snprintf(errmsg, errlen, "%s", msg); for (int c = 0; c < n; c++) cleanup(T[c], 0, false); return err; }
    *ms_per_rep = ms / reps;
    for (int c = 0; c < n; c++) results[c] = T[c].made[results[c]];
    return 0;
}

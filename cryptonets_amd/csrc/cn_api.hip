// libcnhip.so host runtime: contexts, device buffers, launch logic behind the C ABI of include/cnhip.h.
// The compute path is HIP-only: there is no CPU fallback; every entry point fails with CN_ERR_NODEV /
// CN_ERR_HIP when no gfx950 device is usable.
#include "cn_runtime.h"
#include <thread>
#include "cn_k_elem.hip.h"
#include <algorithm>
#include <chrono>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define fail cn_fail
static const RrOps *const rr_ops[3] = {&cn_rr_u64, &cn_rr_f64, &cn_rr_f64l};
static const KsOps *const ks_ops[3] = {&cn_ks_u64, &cn_ks_f64, &cn_ks_f64l};

// ---------------------------------------------------------------- helpers
// deferred submission of per-ciphertext calls (second half of this file)
static DeferQueue *cn_defer_new();
static void cn_defer_delete(DeferQueue *q);
static int cn_defer_flush(cn_ctx *ctx);
static bool cn_defer_pending(cn_ctx *ctx);
enum { DOP_GEMM1 = 0, DOP_ADD, DOP_SUB, DOP_ADDPLAIN, DOP_SUBPLAIN, DOP_MULRELIN, DOP_ENCRYPT,
       DOP_COPY, DOP_MULPLAIN, DOP_ROT, DOP_ROTADD, DOP_COLS, DOP_COLSADD, DOP_SUMSLOTS, DOP_TYPES };      // DOP_COPY .. : staged (gather / batched call / scatter) at flush time
struct DOp {
    int type; int32_t level;
    uint64_t *out;                 // output ciphertext (size 2)
    const uint64_t *a, *b;         // operands (ADD/SUB/MULRELIN: ciphertexts; ADDPLAIN/SUBPLAIN: b = plaintext polynomial)
    uint32_t K; size_t terms;      // GEMM1: K (address, weight) pairs from DeferQueue::addr / ::wt [terms ..)
    const uint64_t *bias;          // GEMM1: plaintext polynomial added to the result (an AddPlain folded in at flush time), or null
    uint64_t nonce = 0, item = 0;  // ENCRYPT: the call's seed and the sampler item of this ciphertext (a = plaintext polynomial or null)
    int64_t arg = 0;               // staged kinds: rotation steps (ROT, ROTADD) / slot count (SUMSLOTS); MULPLAIN: b = plaintext polynomial; ROTADD / COLSADD: b = accumulator
    int32_t fold_first = -1; uint32_t fold_count = 0;   // GEMM1: terms [fold_first, +fold_count) of DeferQueue::folds - zero encryptions folded onto this output (cn_defer_flush)
};
struct DeferQueue {
    std::vector<DOp> ops;
    std::vector<uint64_t> addr, wt;
    struct Haz { int32_t w = -1, r = -1, wop = -1; uint32_t readers = 0; int32_t hd = 0; };   // level of the last writer / deepest reader since / index of the writing op / readers since / heavy depth of the value (defer_push)
    // address -> hazard record: open addressing, cleared by bumping the epoch (a dense-layer call touches 845 records)
    struct HazMap {
        struct E { const uint64_t *key = nullptr; uint32_t epoch = 0; Haz v; };
        std::vector<E> tab = std::vector<E>(1 << 12);
        uint32_t epoch = 1; size_t used = 0;
        static size_t hash(const uint64_t *p) { uint64_t x = (uint64_t)p >> 8; x *= 0x9E3779B97F4A7C15ull; return (size_t)(x >> 20); }
        Haz *find(const uint64_t *p) {
            for (size_t i = hash(p) & (tab.size() - 1);; i = (i + 1) & (tab.size() - 1)) {
                if (tab[i].epoch != epoch) return nullptr;
                if (tab[i].key == p) return &tab[i].v;
            }
        }
        Haz &operator[](const uint64_t *p) {
            if (2 * (used + 1) > tab.size()) grow();
            for (size_t i = hash(p) & (tab.size() - 1);; i = (i + 1) & (tab.size() - 1)) {
                if (tab[i].epoch != epoch) { tab[i].key = p; tab[i].epoch = epoch; tab[i].v = Haz(); used++; return tab[i].v; }
                if (tab[i].key == p) return tab[i].v;
            }
        }
        void grow() {
            std::vector<E> old; old.swap(tab);
            tab.assign(old.size() * 2, E()); used = 0;
            for (const E &e : old) if (e.epoch == epoch) (*this)[e.key] = e.v;
        }
        void clear() { used = 0; if (++epoch == 0) { for (E &e : tab) e.epoch = 0; epoch = 1; } }
    } haz;
    int32_t maxlevel = -1;
    std::vector<std::pair<uint64_t *, size_t>> frees;      // arrays released by the caller while calls were pending: back to the pool after the flush
    struct Fold { int32_t enc; uint64_t w; };               // a zero encryption (index of its DOp) folded into a scalar product with weight w (residue mod t)
    std::vector<Fold> folds;
};
// Small arrays (a per-ciphertext caller allocates every Ciphertext on its own: thousands of 640 KiB arrays per layer) are carved out of
// slabs - one hipMalloc per SLAB_PIECES arrays, neighbours in the address space - and only ever travel between the handles and the pool;
// the slabs themselves are released with the context.
static const size_t SLAB_MAX_ITEM = 8u << 20, SLAB_BYTES = 64u << 20;
struct Slab { char *base; size_t bytes; };
static std::vector<Slab> &slabs_of(cn_ctx *ctx) { return *reinterpret_cast<std::vector<Slab> *>(ctx->slabs); }
static bool in_slab(cn_ctx *ctx, const void *p) {
    for (const Slab &s : slabs_of(ctx)) if ((const char *)p >= s.base && (const char *)p < s.base + s.bytes) return true;
    return false;
}
static bool deferring(cn_ctx *ctx);
static bool zero_fold_ok(cn_ctx *ctx);
static int flush_zero_folds(cn_ctx *ctx, DeferQueue *q, const std::vector<const DOp *> &gemms);
static int defer_staged(cn_ctx *ctx, int type, Buffer *A, uint32_t ai, Buffer *B, uint32_t bi, const uint64_t *plain, uint32_t pstride_words, Buffer *O, uint32_t oi,
                        uint32_t count, int64_t arg);
static const uint32_t DEFER_STAGED_MAX = 4;       // per-ciphertext callers: calls on up to this many ciphertexts are queued, larger ones run at once
static void pool_flush(cn_ctx *ctx);
static int free_gemm_plan(cn_ctx *ctx, Buffer &b);
static int free_graph(cn_ctx *ctx, Buffer &b);
static int use(cn_ctx *c) { HIPCHK(hipSetDevice(c->device)); return 0; }

static int ensure_scratch(cn_ctx *c, size_t bytes) {
    c->soff = 0;
    if (bytes <= c->scap) return 0;
    if (c->capturing || c->graphs_alive) return fail(CN_ERR_ARG, "the scratch arena would have to grow while a graph is recorded / alive: run the sequence once before cn_graph_begin");
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->scratch) HIPCHK(hipFree(c->scratch));
    c->scratch = nullptr; c->scap = 0;
    size_t want = bytes + (bytes >> 3) + (1 << 20);
    if (hipMalloc((void **)&c->scratch, want) != hipSuccess) {      // HBM full: return the cached handle arrays and retry once
        (void)hipGetLastError();
        pool_flush(c);
        HIPCHK(hipMalloc((void **)&c->scratch, want));
    }
    c->scap = want;
    return 0;
}
static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
template <class T> static T *salloc(cn_ctx *c, size_t count) {
    size_t b = al(count * sizeof(T));
    if (c->soff + b > c->scap) return nullptr;
    T *p = (T *)(c->scratch + c->soff); c->soff += b; return p;
}
// Small host tables (gather indices, weight tiles) travel with an asynchronous copy on the context stream.  The caller's buffer is
// usually a local std::vector, so the bytes are first moved into a staging block the context keeps alive until the stream has
// drained (checked lazily) - the copy never reads memory that has gone out of scope, whatever the runtime does with pageable sources.
// Host side of the small uploads: a ring of PINNED memory per context.  hipMemcpyAsync from pageable memory is staged by the runtime and
// holds the calling thread for ~10 us a piece; a flush of queued LoLa calls uploads a few hundred small tables.  A block of the ring is
// reused only after the stream has passed it: the ring synchronises once per lap.
static char *pin_block(cn_ctx *c, size_t bytes) {
    const size_t cap = 8u << 20;
    bytes = (bytes + 63) & ~(size_t)63;
    if (bytes > cap / 2) return nullptr;
    if (!c->pin) { if (hipHostMalloc((void **)&c->pin, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->pin = nullptr; return nullptr; } c->pin_off = 0; }
    if (c->pin_off + bytes > cap) { if (hipStreamSynchronize(c->stream) != hipSuccess) return nullptr; c->pin_off = 0; }
    char *p = c->pin + c->pin_off;
    c->pin_off += bytes;
    return p;
}
static int upload_bytes(cn_ctx *c, const void *host, size_t bytes, void *dev) {
    if (!c->capturing) {
        if (char *p = pin_block(c, bytes)) {
            memcpy(p, host, bytes);
            HIPCHK(hipMemcpyAsync(dev, p, bytes, hipMemcpyHostToDevice, c->stream));
            return 0;
        }
    }
    std::vector<std::unique_ptr<char[]>> &keep = c->capturing ? c->cap_staged : c->staged;      // a recorded upload reads its host block at EVERY launch
    if (!c->capturing && !keep.empty() && hipStreamQuery(c->stream) == hipSuccess) keep.clear();
    (void)hipGetLastError();                                   // hipStreamQuery reports hipErrorNotReady through the sticky error as well
    keep.emplace_back(new char[bytes]);
    memcpy(keep.back().get(), host, bytes);
    HIPCHK(hipMemcpyAsync(dev, keep.back().get(), bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}
// A SMALL table (a few hundred bytes of operand addresses) does not travel at all: the kernels read it where pin_block put it - pinned host memory is
// mapped into the device's address space.  A copy would be one more dispatch in front of the kernel that needs the table (the runtime's blit kernel),
// and dependent dispatches are what the latency of a single-image chain is made of (20 of the 125 dispatches of a LoLa image were
// such copies).  *dev = where the kernel finds the table: inside the pinned ring, or `fallback` (device memory the caller owns) after a copy when the
// table is large, a graph is being recorded (a recorded launch must find its table at every replay) or CN_TABLES_ZERO_COPY=0.
static const size_t SMALL_TABLE_BYTES = 4096;
static int place_table(cn_ctx *c, const void *host, size_t bytes, void *fallback, const void **dev) {
    static const bool zero_copy = !(getenv("CN_TABLES_ZERO_COPY") && !atoi(getenv("CN_TABLES_ZERO_COPY")));
    if (zero_copy && !c->capturing && bytes <= SMALL_TABLE_BYTES) {
        if (char *p = pin_block(c, bytes)) {
            if (!c->pin_dev) { void *d = nullptr; if (hipHostGetDevicePointer(&d, c->pin, 0) == hipSuccess) c->pin_dev = (char *)d; else (void)hipGetLastError(); }
            if (c->pin_dev) { memcpy(p, host, bytes); *dev = c->pin_dev + (p - c->pin); return 0; }
        }
    }
    *dev = fallback;
    return upload_bytes(c, host, bytes, fallback);
}
template <class T> static int upload_tmp(cn_ctx *c, const T *host, size_t count, T **dev) {
    *dev = salloc<T>(c, count);
    if (!*dev) return fail(CN_ERR_HIP, "internal: scratch exhausted");
    return upload_bytes(c, host, count * sizeof(T), *dev);
}
static Buffer *getbuf(cn_ctx *c, cn_handle h, int kind) {
    Buffer *b = c->bufs.find(h);
    return (b && b->kind == kind) ? b : nullptr;
}
static int range_ok(const Buffer *b, uint32_t first, uint32_t count, uint32_t stride = 1) {
    if (!count) return 1;
    uint64_t last = (uint64_t)first + (uint64_t)(count - 1) * stride;
    return last < b->count;
}
#define GETCT(var, h, sz) Buffer *var = getbuf(ctx, h, 0); if (!var) return fail(CN_ERR_ARG, "invalid ciphertext handle " #h); \
    if ((sz) && var->size != (uint32_t)(sz)) return fail(CN_ERR_ARG, "ciphertext size mismatch for " #h)
#define GETPT(var, h) Buffer *var = getbuf(ctx, h, 1); if (!var) return fail(CN_ERR_ARG, "invalid plaintext handle " #h)
// every entry point takes the context lock; all but the deferrable ones (cn_defer.hip) first drain the queue of deferred calls
// (the function bodies that start with LOCK / LOCK_ONLY run inside CnMutex::run - see API_BODY below: under the context lock, on the calling
// thread or, when the lock is held, on the holder's thread)
// Lock-free submission ("defer" = 2, cn_submit.h): every entry point that takes the lock first executes the records other threads have published up to
// this moment (ring_sync: in claim order, waiting for a slot that is claimed but not yet written); LOCK additionally reports the first error one of them ran into.
static int ring_sync(cn_ctx *ctx, bool report);
#define LOCK_ONLY CHECK(use(ctx)); CHECK(ring_sync(ctx, false))
#define API_BODY return ctx->mu.run([&]() -> int {
#define API_END });
#define LOCK CHECK(use(ctx)); CHECK(ring_sync(ctx, true)); CHECK(cn_defer_flush(ctx))

#define launch_count cn_launch_count

int cn_run_ntt(cn_ctx *c, uint64_t *data, uint32_t limbs, uint32_t base_off, uint32_t nmod, int inverse) {
    if (!limbs) return 0;
    uint32_t n = c->hc.n;
    bool f64 = c->use_f64;
    for (uint32_t m = base_off; m < base_off + nmod; m++) f64 = f64 && c->hc.f64ok[m];
    bool light = f64;
    for (uint32_t m = base_off; m < base_off + nmod && light; m++) {
        uint64_t q = m < c->hc.k ? c->hc.q[m].q : (m < c->hc.k + c->hc.kb ? c->hc.bsk[m - c->hc.k].q : c->hc.t.q);
        if (q >> 44) light = false;
    }
    bool done = !c->legacy_ntt && rr_ops[light ? POL_F64L : (f64 ? POL_F64 : POL_U64)]->ntt(c, data, limbs, base_off, nmod, inverse);
    if (!done) {
        uint32_t nt = std::min<uint32_t>(512, n / 2);
        hipLaunchKernelGGL(k_ntt, dim3(limbs), dim3(nt), (size_t)n * 8, c->stream, data, c->dc, base_off, nmod, inverse);
    }
    HIPCHK(hipGetLastError());
    launch_count(c);
    if (inverse) c->st.ntt_inverse_limbs += limbs; else c->st.ntt_forward_limbs += limbs;
    return 0;
}

// ---------------------------------------------------------------- misc API
extern "C" int cn_version(void) { return 100; }
extern "C" int cn_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
extern "C" int cn_default_coeff_modulus(uint32_t n, uint64_t *q) { return cn_default_coeff_modulus_impl(n, q); }

template <class K> static int big_lds(K kern, size_t bytes) {
    HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
template <int EPT> static int set_ks_attr(size_t bytes) {
    HIPCHK(hipFuncSetAttribute((const void *)k_keyswitch<EPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

// ---- a HARDWARE queue of its own for every context.  HIP deals the streams of a process onto at most GPU_MAX_HW_QUEUES (4) hardware queues, the
// null stream and the runtime's transfer queue take part, and which streams end up together depends on the order in which the process happened to
// create them.  Two contexts whose streams share a queue run their kernels one after the other: the four plaintext-prime channels of one LoLa
// inference took 10.0-10.7 ms per image in most processes and 7.5 ms in those where the four streams happened to sit on four queues
// (profiles/r03_stream_queues.txt; raising GPU_MAX_HW_QUEUES to 8 is no way out: 12 ms).  The runtime offers no query, so the library measures:
// two 100 us spin kernels, one per stream - together they take 100 us on two queues and 200 us on one.  A new context keeps the first stream
// that overlaps with the streams of every live context on its device; rejected candidates stay alive until the choice is made (the runtime
// hands a new stream the least-used queue) and are destroyed then.  CN_STREAM_PROBE=0 takes the first stream as it comes.
__global__ void k_spin(uint64_t ticks) {
    const uint64_t t0 = wall_clock64();                        // constant 100 MHz counter
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static std::mutex g_ctx_reg_mu;
static std::vector<cn_ctx *> g_ctx_reg;
// contexts of ONE device are created one after the other (the selection below needs to see every live stream of the device); other devices' creations and
// every destruction go on meanwhile (round 5, ADVICE r04: the probe used to run under the registry lock - a busy context stalled every create / destroy in the process)
static std::mutex &device_create_mutex(int device) { static std::mutex mu[64]; return mu[(unsigned)device % 64]; }
// (both streams idle and nobody else submitting to them: the caller holds the lock of the context that owns `b`)
static bool streams_share_a_queue(hipStream_t a, hipStream_t b) {
    double best = 1e9;
    for (int rep = 0; rep < 3 && best > 150e-6; rep++) {
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return false; }   // (sticky error cleared)
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, (uint64_t)10000);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, (uint64_t)10000);
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return false; }
        best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    return best > 150e-6;
}
// Caller holds device_create_mutex(c->device).  The contexts to probe are SNAPSHOT under the registry lock and pinned (cn_ctx::probe_pins: cn_ctx_destroy of a
// pinned context waits for the probe to let go); the registry lock is released before any stream is touched.  The stream of another live context is only touched UNDER THAT
// CONTEXT'S LOCK, with `capturing` re-read under it: no API call of another thread can submit to it, start recording on it or free it while the two
// spin kernels run, and a recording stream never sees a foreign launch (the probe used to read o->capturing and launch on o->stream
// unsynchronised).  Lock order: device-create mutex, then (briefly) the registry, then ONE context lock at a time; nothing takes them the other way round
// (cn_ctx_destroy leaves the registry lock before it takes the context's).  A probe waits for the work the other context has queued (at most a few batches); CN_STREAM_PROBE=0
// switches the whole selection off.  No early return between the creation of a candidate and the clean-up below: rejected candidates are destroyed on
// every path.
static int pick_stream(cn_ctx *c) {
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->stream_tries = 1;
    const char *env = getenv("CN_STREAM_PROBE");
    if (env && !atoi(env)) return 0;
    std::vector<cn_ctx *> others;
    {
        std::lock_guard<std::mutex> reg(g_ctx_reg_mu);
        for (cn_ctx *o : g_ctx_reg) if (o->device == c->device) { o->probe_pins.fetch_add(1, std::memory_order_acq_rel); others.push_back(o); }
    }
    struct Unpin { std::vector<cn_ctx *> &v; ~Unpin() { for (cn_ctx *o : v) o->probe_pins.fetch_sub(1, std::memory_order_acq_rel); } } unpin{others};
    if (others.empty()) return 0;
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, c->stream, (uint64_t)1);          // code object loaded, queue created
    if (hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); return 0; }          // (no selection; the context itself will report a broken device)
    auto collides = [&](hipStream_t s) {
        for (cn_ctx *o : others) {
            CnGuard lk(o->mu);
            if (o->capturing) continue;                       // a recording stream must not see foreign launches: not probed
            if (streams_share_a_queue(s, o->stream)) return true;
        }
        return false;
    };
    std::vector<hipStream_t> rejected;
    const hipStream_t first = c->stream;
    bool found = !collides(c->stream);
    while (!found && c->stream_tries < 6) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        rejected.push_back(c->stream);
        c->stream = s;
        c->stream_tries++;
        found = !collides(c->stream);
    }
    if (!found) { rejected.push_back(c->stream); c->stream = first; c->stream_tries = -c->stream_tries; }     // more contexts than queues: as it came
    for (hipStream_t s : rejected) if (s != c->stream) (void)hipStreamDestroy(s);
    return 0;
}

static int ctx_init(cn_ctx *c, uint32_t n, uint32_t k, int device, std::vector<uint64_t> &tw);
static void ctx_teardown(cn_ctx *ctx);
extern "C" int cn_ctx_create(uint32_t n, const uint64_t *q, uint32_t k, uint64_t t, int dbc, int gdbc, int device, cn_ctx **out) {
    if (!out || !q) return fail(CN_ERR_ARG, "null argument");
    int ndev = cn_device_count();
    if (ndev <= 0) return fail(CN_ERR_NODEV, "no HIP device available (libcnhip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(CN_ERR_ARG, "device %d out of range (%d devices)", device, ndev);
    if (n > 16384) return fail(CN_ERR_ARG, "poly modulus degree %u too large for the LDS-resident NTT (max 16384)", n);
    if (k == 0 || k > CN_MAXK) return fail(CN_ERR_ARG, "coeff modulus count out of range");
    std::vector<uint64_t> tw((size_t)(2 * k + 3) * 4 * n);             // k data + up to k + 2 auxiliary moduli + t
    cn_ctx *c = new cn_ctx();
    char err[256];
    c->index_map.assign(n, 0);
    if (cn_build_consts(&c->hc, n, q, k, t, dbc, gdbc, tw.data(), c->index_map.data(), err, sizeof err)) { delete c; return fail(CN_ERR_ARG, "%s", err); }
    c->dq = cn_defer_new();
    c->ring = new SubmitRing(); c->ready = new ReadyRing();
    c->slabs = new std::vector<Slab>();
    std::lock_guard<std::mutex> one_at_a_time(device_create_mutex(device));       // until this context is registered: the next creator on this device must see its stream
    const int rc = ctx_init(c, n, k, device, tw);
    if (rc) {                                   // whatever was created so far is released (streams, events, tables); the message of the failing call stays
        (void)hipGetLastError();
        ctx_teardown(c);
        return rc;
    }
    { std::lock_guard<std::mutex> reg(g_ctx_reg_mu); g_ctx_reg.push_back(c); }
    *out = c;
    return 0;
}
static int ctx_init(cn_ctx *c, uint32_t n, uint32_t k, int device, std::vector<uint64_t> &tw) {
    c->device = device;
    HIPCHK(hipSetDevice(device));
    CHECK(pick_stream(c));
    HIPCHK(hipEventCreate(&c->ev0)); HIPCHK(hipEventCreate(&c->ev1));
    HIPCHK(hipMalloc((void **)&c->tw, tw.size() * 8));
    HIPCHK(hipMemcpy(c->tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    c->hc.tw = c->tw;
    {
        std::vector<double> twd((size_t)(2 * k + 3) * 2 * n, 0.0);
        cn_build_f64_tables(&c->hc, tw.data(), twd.data());
        HIPCHK(hipMalloc((void **)&c->twd, twd.size() * 8));
        HIPCHK(hipMemcpy(c->twd, twd.data(), twd.size() * 8, hipMemcpyHostToDevice));
        c->hc.twd = c->twd;
        c->hc.twdh = nullptr;
        if (c->hc.logn == 14 && c->hc.q_f64) {                  // N = 16384: half tables of the split key switch
            std::vector<double> th((size_t)k * 4 * (n / 2), 0.0);
            cn_build_half_tables(&c->hc, tw.data(), th.data());
            HIPCHK(hipMalloc((void **)&c->twdh, th.size() * 8));
            HIPCHK(hipMemcpy(c->twdh, th.data(), th.size() * 8, hipMemcpyHostToDevice));
            c->hc.twdh = c->twdh;
        }
        c->use_f64 = !(getenv("CN_NO_F64") && atoi(getenv("CN_NO_F64")));
    }
    HIPCHK(hipMalloc((void **)&c->dc, sizeof(DevConsts)));
    HIPCHK(hipMemcpy(c->dc, &c->hc, sizeof(DevConsts), hipMemcpyHostToDevice));
    c->bs = std::min<uint32_t>(256, n); c->chunks = n / c->bs;
    c->ctw2 = (size_t)2 * k * n;
    const char *env = getenv("CN_SCRATCH_GB");
    c->smax = (size_t)((env ? atof(env) : 24.0) * (double)(1ull << 30));
    env = getenv("CN_POOL_GB");
    c->pool_max = (size_t)((env ? atof(env) : 8.0) * (double)(1ull << 30));
    c->legacy_ntt = getenv("CN_LEGACY_NTT") && atoi(getenv("CN_LEGACY_NTT"));
    c->ks_tight = getenv("CN_KS_TIGHT") && atoi(getenv("CN_KS_TIGHT"));
    // fused key switch, workgroup order: limb-major up to N = 8192 (one key slice per XCD L2 at a time: 2.34 GiB fetched per 845-ciphertext launch
    // against 3.99 GiB in (ciphertext, limb) order and 3.0 GiB with the limbs of a ciphertext on one XCD, same kernel time -
    // profiles/r03_pmc_keyswitch_orders.txt); (ciphertext, limb) order at N = 16384, where limb-major measured 30 % slower in round 1
    c->ks_xcd = c->hc.logn <= 13 ? 2 : 1;            // N = 16384 (k_keyswitch_pair14): the k workgroups of a ciphertext on one XCD - they share its source limbs in that L2 (34.3 vs 35.1 ms per 5488-ciphertext link)
    if (getenv("CN_KS_XCD")) c->ks_xcd = atoi(getenv("CN_KS_XCD"));
    if (getenv("CN_KS_PAIR14")) c->ks_pair14 = atoi(getenv("CN_KS_PAIR14")) != 0;             // A/B switches of the N = 16384 key switch (round 5)
    if (getenv("CN_KS_CHAIN")) c->ks_chain = atoi(getenv("CN_KS_CHAIN")) != 0;
    if (getenv("CN_SQ_FUSED")) c->sq_fused = atoi(getenv("CN_SQ_FUSED")) != 0;
    if (getenv("CN_SQ_LDS")) c->sq_lds = atoi(getenv("CN_SQ_LDS")) != 0;
    if (getenv("CN_SQ_PIPE")) c->sq_pipe = atoi(getenv("CN_SQ_PIPE"));
    if (getenv("CN_SQ_OVERLAP")) c->sq_overlap = atoi(getenv("CN_SQ_OVERLAP")) != 0;
    if (getenv("CN_ENC_FUSED")) c->enc_fused = atoi(getenv("CN_ENC_FUSED")) != 0;
    if (getenv("CN_FOLD_ZERO")) c->fold_zero = atoi(getenv("CN_FOLD_ZERO")) != 0;
    HIPCHK(hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device));
    if (getenv("CN_GEMM_MFMA")) c->gemm_mfma = atoi(getenv("CN_GEMM_MFMA")) != 0;
    if (getenv("CN_GEMM_PAIR")) c->gemm_pair = atoi(getenv("CN_GEMM_PAIR")) != 0;
    if (getenv("CN_GEMM_ORDER")) c->gemm_order = atoi(getenv("CN_GEMM_ORDER"));
    if (getenv("CN_MP_FUSED")) c->mp_fused = atoi(getenv("CN_MP_FUSED")) != 0;          // A/B switch of the fused squaring kernel
    size_t lds = (size_t)ntt_lds_words(n) * 8;
    if (lds > 48 * 1024) {                 // N >= 8192: the padded LDS image exceeds the default dynamic-LDS limit
        CHECK(big_lds(k_ntt, lds)); CHECK(big_lds(k_galois_lds, (size_t)n * 8)); CHECK(big_lds(k_galois_limbs, (size_t)n * 8));
        CHECK(set_ks_attr<8>(lds)); CHECK(set_ks_attr<16>(lds));
    }
    for (int pol = 0; pol < 3; pol++) { CHECK(rr_ops[pol]->set_attrs(c->hc.logn, lds)); CHECK(ks_ops[pol]->set_attrs(c->hc.logn, lds)); }
    return 0;
}
extern "C" int cn_ctx_destroy(cn_ctx *ctx) {
    if (!ctx) return 0;
    {
        std::lock_guard<std::mutex> reg(g_ctx_reg_mu);
        g_ctx_reg.erase(std::remove(g_ctx_reg.begin(), g_ctx_reg.end(), ctx), g_ctx_reg.end());
    }
    while (ctx->probe_pins.load(std::memory_order_acquire) > 0) std::this_thread::yield();      // a cn_ctx_create on this device is measuring this context's stream (pick_stream)
    ctx_teardown(ctx);
    return 0;
}
// releases everything a context owns; also the clean-up of a cn_ctx_create that failed half way (every member is null / empty until it is created)
static void ctx_teardown(cn_ctx *ctx) {
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {   // queued per-ciphertext calls are launched (their results die with the context, but the arrays parked behind them - cn_free while
        // calls were pending - go back to the pool and are released with it)
        CnGuard lk(ctx->mu);
        if (ctx->capturing) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(ctx->stream, &g); if (g) (void)hipGraphDestroy(g); ctx->capturing = false; }
        (void)ring_sync(ctx, false);
        (void)cn_defer_flush(ctx);
    }
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->bufs.for_each([&](Buffer &b) {
        if (b.kind == 2) (void)free_gemm_plan(ctx, b);
        else if (b.kind == 3) (void)free_graph(ctx, b);
        else if (!in_slab(ctx, b.d)) (void)hipFree(b.d);
    });
    pool_flush(ctx);
    for (const Slab &sl : slabs_of(ctx)) (void)hipFree(sl.base);
    delete &slabs_of(ctx);
    if (ctx->rlk.owned) (void)hipFree(ctx->rlk.d);
    for (auto &kv : ctx->gk) if (kv.second.owned) (void)hipFree(kv.second.d);
    (void)hipFree(ctx->sk); (void)hipFree(ctx->pk); (void)hipFree(ctx->ks_part); (void)hipFree(ctx->d_index_map); (void)hipFree(ctx->stage); if (ctx->pin) (void)hipHostFree(ctx->pin);
    (void)hipFree(ctx->scratch); (void)hipFree(ctx->tw); (void)hipFree(ctx->twd); (void)hipFree(ctx->twdh); (void)hipFree(ctx->dc);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    cn_defer_delete(ctx->dq);
    delete ctx->ring; delete ctx->ready;
    delete ctx;
}
static int free_body(cn_ctx *ctx, cn_handle h);
extern "C" int cn_set_option(cn_ctx *ctx, const char *name, int value) { API_BODY
    LOCK;
    if (!name) return fail(CN_ERR_ARG, "null option name");
    if (!strcmp(name, "f64")) { ctx->use_f64 = value != 0; return 0; }              // affects keys uploaded AFTER the call
    if (!strcmp(name, "legacy_ntt")) { ctx->legacy_ntt = value != 0; return 0; }
    if (!strcmp(name, "ks_tight")) { ctx->ks_tight = value != 0; return 0; }
    if (!strcmp(name, "gemm_order")) { ctx->gemm_order = value; return 0; }                  // 1 slice-major (default), 0 group-major
    if (!strcmp(name, "ks_perm_fused")) { ctx->ks_perm_fused = value != 0; return 0; }      // rotations of small batches: automorphism inside the key-switch kernels (default 1)
    if (!strcmp(name, "ks_xcd")) { ctx->ks_xcd = value; return 0; }              // 0 (ct, limb) order, 1 the limbs of a ciphertext on one XCD, 2 limb-major
    if (!strcmp(name, "sq_fused")) { ctx->sq_fused = value != 0; return 0; }
    if (!strcmp(name, "sq_lds")) { ctx->sq_lds = value != 0; return 0; }
    if (!strcmp(name, "sq_pipe")) { ctx->sq_pipe = value; return 0; }
    if (!strcmp(name, "sq_overlap")) { ctx->sq_overlap = value != 0; return 0; }
    if (!strcmp(name, "enc_fused")) { ctx->enc_fused = value != 0; return 0; }
    if (!strcmp(name, "fold_zero")) { ctx->fold_zero = value != 0; return 0; }      // queued zero encryptions that only feed a queued scalar product: folded by linearity (default 1)
    if (!strcmp(name, "gemm_mfma")) { ctx->gemm_mfma = value != 0; return 0; }        // affects GEMMs planned AFTER the call
    if (!strcmp(name, "gemm_pair")) { ctx->gemm_pair = value != 0; return 0; }        // likewise
    if (!strcmp(name, "mp_fused")) { ctx->mp_fused = value != 0; return 0; }
    if (!strcmp(name, "ks_wide")) { ctx->ks_wide = value; return 0; }
    if (!strcmp(name, "ks_split14")) { ctx->ks_split14 = value != 0; return 0; }
    if (!strcmp(name, "ks_pair14")) { ctx->ks_pair14 = value != 0; return 0; }
    if (!strcmp(name, "ks_chain")) { ctx->ks_chain = value != 0; return 0; }
    if (!strcmp(name, "mp_bcast")) { ctx->mp_bcast = value != 0; return 0; }
    if (!strcmp(name, "defer")) {                // 0 immediate, 1 queued under the context lock, 2 queued through the lock-free submission ring; the queue was drained by LOCK
        if (value < 0 || value > 2) return fail(CN_ERR_ARG, "defer: 0, 1 or 2");
        ctx->defer.store(value, std::memory_order_release);
        if (value != 2) {                        // the ready single-ciphertext arrays of the lock-free mode go back to the pool
            while (const cn_handle h = ctx->ready->pop()) CHECK(free_body(ctx, h));
            ctx->ready->misses.store(0, std::memory_order_relaxed);
        }
        return 0;
    }
    if (!strcmp(name, "ks_xi")) {                // decomposition convention of the key switch (DevConsts::ks_xi); the keys must be of the same convention
        if (ctx->capturing || ctx->graphs_alive) return fail(CN_ERR_ARG, "ks_xi cannot change while a graph is recorded or alive (its kernels were chosen for the other convention)");
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->hc.ks_xi = value != 0;
        HIPCHK(hipMemcpy(ctx->dc, &ctx->hc, sizeof(DevConsts), hipMemcpyHostToDevice));
        return 0;
    }
    return fail(CN_ERR_ARG, "unknown option %s", name);
API_END }
// read-back of the switches and of choices the library made (tests, diagnostics)
extern "C" int cn_get_option(cn_ctx *ctx, const char *name, int *value) { API_BODY
    LOCK_ONLY;
    if (!name || !value) return fail(CN_ERR_ARG, "null argument");
    if (!strcmp(name, "f64")) *value = ctx->use_f64;
    else if (!strcmp(name, "defer")) *value = ctx->defer.load(std::memory_order_relaxed);
    else if (!strcmp(name, "ready_handles")) *value = (int)ctx->ready->size();          // allocated single-ciphertext arrays waiting for a lock-free cn_ct_alloc
    else if (!strcmp(name, "ks_wide")) *value = ctx->ks_wide;
    else if (!strcmp(name, "ks_xi")) *value = (int)ctx->hc.ks_xi;
    else if (!strcmp(name, "ks_xcd")) *value = ctx->ks_xcd;
    else if (!strcmp(name, "ks_pair14")) *value = ctx->ks_pair14;
    else if (!strcmp(name, "ks_chain")) *value = ctx->ks_chain;
    else if (!strcmp(name, "mp_bcast")) *value = ctx->mp_bcast;
    else if (!strcmp(name, "sq_fused")) *value = ctx->sq_fused;
    else if (!strcmp(name, "mp_fused")) *value = ctx->mp_fused;
    else if (!strcmp(name, "gemm_mfma")) *value = ctx->gemm_mfma;
    else if (!strcmp(name, "gemm_pair")) *value = ctx->gemm_pair;
    else if (!strcmp(name, "sq_lds")) *value = ctx->sq_lds;
    else if (!strcmp(name, "sq_pipe")) *value = ctx->sq_pipe;
    else if (!strcmp(name, "sq_overlap")) *value = ctx->sq_overlap;
    else if (!strcmp(name, "enc_fused")) *value = ctx->enc_fused;
    else if (!strcmp(name, "fold_zero")) *value = ctx->fold_zero;
    else if (!strcmp(name, "folded_zero_encryptions")) *value = (int)std::min<uint64_t>(ctx->folded_zero, 0x7fffffff);    // zero encryptions folded so far (tests)
    else if (!strcmp(name, "behz_small_base")) *value = ctx->hc.bsk[ctx->hc.kb - 1].q < (1ull << 49);     // auxiliary primes below 2^49 (FP64 kernels) instead of SEAL's 61-bit ones
    else if (!strcmp(name, "behz_f64")) *value = ctx->hc.behz_f64 && ctx->use_f64;
    else if (!strcmp(name, "aux_primes")) *value = (int)ctx->hc.kb;
    else if (!strcmp(name, "pending_calls")) *value = (int)ctx->dq->ops.size();
    else if (!strcmp(name, "ks_perm_fused")) *value = ctx->ks_perm_fused;
    else if (!strcmp(name, "gemm_order")) *value = ctx->gemm_order;
    else if (!strcmp(name, "stream_tries")) *value = ctx->stream_tries;           // streams created until one had a hardware queue of its own (< 0: none had)
    else return fail(CN_ERR_ARG, "unknown option %s", name);
    return 0;
API_END }
#define NOT_CAPTURING(what) do { if (ctx->capturing) return fail(CN_ERR_ARG, what " is not possible while a graph is recorded (cn_graph_begin .. cn_graph_end)"); } while (0)
extern "C" int cn_sync(cn_ctx *ctx) { API_BODY LOCK; NOT_CAPTURING("cn_sync"); HIPCHK(hipStreamSynchronize(ctx->stream)); ctx->staged.clear(); return 0; API_END }
extern "C" void *cn_stream(cn_ctx *ctx) { return (void *)ctx->stream; }
// ctx's later work waits (on the device) for other's earlier work.  The two locks are taken one after the other, never together.
extern "C" int cn_ctx_wait_for(cn_ctx *ctx_, cn_ctx *other) {
    if (!ctx_ || !other) return fail(CN_ERR_ARG, "null argument");
    if (ctx_ == other) return 0;
    hipEvent_t ev = nullptr;
    {
        cn_ctx *ctx = other;                                  // (the macros name the context `ctx`)
        const int rc = ctx->mu.run([&]() -> int {
            LOCK; NOT_CAPTURING("cn_ctx_wait_for");
            if (!ctx->ev_order) HIPCHK(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
            HIPCHK(hipEventRecord(ctx->ev_order, ctx->stream)); ev = ctx->ev_order;
            return 0;
        });
        if (rc) return rc;
    }
    cn_ctx *ctx = ctx_;
    return ctx->mu.run([&]() -> int {
        LOCK; NOT_CAPTURING("cn_ctx_wait_for");
        HIPCHK(hipStreamWaitEvent(ctx->stream, ev, 0));
        return 0;
    });
}
extern "C" size_t cn_key_words(cn_ctx *ctx, int which) { return (size_t)(which ? ctx->hc.gk_tot : ctx->hc.rl_tot) * ctx->ctw2; }

// does this context keep its key-switch keys as FP64 images (the FP64 key-switch kernels read doubles)?
static bool keys_as_f64(const cn_ctx *ctx) { return ctx->use_f64 && ctx->hc.q_f64 && !ctx->legacy_ntt && ctx->hc.logn >= 10 && ctx->hc.logn <= 14; }
// coeff_form: the words are coefficient-form polynomials [..][k][N]; the device transforms them with its own tables (cn_load_key, form 1)
static int set_key(cn_ctx *ctx, KsKey &slot, const uint64_t *words, size_t count, size_t expect, int is_dev, bool coeff_form = false) {
    if (!words || count != expect) return fail(CN_ERR_ARG, "key has %zu words, expected %zu", count, expect);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (slot.owned && slot.d) HIPCHK(hipFree(slot.d));
    slot = {nullptr, false, false};
    if (is_dev) { slot.d = (uint64_t *)words; slot.owned = false; }
    else {
        HIPCHK(hipMalloc((void **)&slot.d, count * 8));
        slot.owned = true;
        HIPCHK(hipMemcpy(slot.d, words, count * 8, hipMemcpyHostToDevice));
    }
    if (coeff_form) CHECK(cn_run_ntt(ctx, slot.d, (uint32_t)(count / ctx->hc.n), 0, ctx->hc.k, 0));
    if (keys_as_f64(ctx)) {
        // FP64 key-switch kernel reads the key as doubles: convert once, in place (an adopted device buffer is converted too)
        hipLaunchKernelGGL(k_u64_to_f64, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, slot.d, count);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        slot.f64 = true;
    }
    return 0;
}
extern "C" int cn_set_relin_key(cn_ctx *ctx, const uint64_t *words, size_t count, int is_dev) { API_BODY
    LOCK; NOT_CAPTURING("cn_set_relin_key"); return set_key(ctx, ctx->rlk, words, count, cn_key_words(ctx, 0), is_dev);
API_END }
extern "C" int cn_set_galois_key(cn_ctx *ctx, uint64_t elt, const uint64_t *words, size_t count, int is_dev) { API_BODY
    LOCK; NOT_CAPTURING("cn_set_galois_key");
    if (!(elt & 1) || elt >= 2ull * ctx->hc.n) return fail(CN_ERR_ARG, "invalid Galois element");
    return set_key(ctx, ctx->gk[elt], words, count, cn_key_words(ctx, 1), is_dev);
API_END }
extern "C" int cn_has_galois_key(cn_ctx *ctx, uint64_t elt) { CnGuard lk(ctx->mu); auto it = ctx->gk.find(elt); return it != ctx->gk.end() && it->second.d; }
extern "C" uint64_t cn_galois_elt_from_step(cn_ctx *ctx, int steps) {
    uint64_t n = ctx->hc.n, m = 2 * n;
    if (steps == 0) return m - 1;
    uint64_t pos = (uint64_t)std::abs((long)steps);
    if (pos >= n / 2) return 0;
    uint64_t s = steps < 0 ? n / 2 - pos : pos, e = 1;
    while (s--) e = (e * 3) & (m - 1);
    return e;
}

// ---------------------------------------------------------------- buffers
static void pool_flush(cn_ctx *ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->pool) {
        std::vector<uint64_t *> keep;
        for (uint64_t *p : kv.second) { if (in_slab(ctx, p)) keep.push_back(p); else { (void)hipFree(p); ctx->pool_bytes -= kv.first; } }
        kv.second.swap(keep);
    }
}
static int dev_alloc(cn_ctx *ctx, size_t bytes, uint64_t **out) {
    auto it = ctx->pool.find(bytes);
    if (it != ctx->pool.end() && !it->second.empty()) {
        *out = it->second.back(); it->second.pop_back(); ctx->pool_bytes -= bytes;
        if (ctx->capturing) ctx->cap_allocs.emplace_back(*out, bytes);
        return 0;
    }
    if (ctx->capturing) {                                       // relaxed capture mode permits hipMalloc (it does not touch the recording stream);
        if (hipMalloc((void **)out, bytes) != hipSuccess) {     // flushing the pool would synchronise, so no retry here
            (void)hipGetLastError();
            return fail(CN_ERR_HIP, "out of device memory while a graph is recorded (%zu bytes)", bytes);
        }
        ctx->cap_allocs.emplace_back(*out, bytes);
        return 0;
    }
    if (bytes <= SLAB_MAX_ITEM && bytes % 256 == 0) {            // a slab of neighbours: one goes to the caller, the rest into the pool
        const size_t pieces = std::min<size_t>(64, std::max<size_t>(4, SLAB_BYTES / bytes));
        char *base = nullptr;
        if (hipMalloc((void **)&base, pieces * bytes) == hipSuccess) {
            slabs_of(ctx).push_back({base, pieces * bytes});
            std::vector<uint64_t *> &pl = ctx->pool[bytes];
            for (size_t i = pieces; i-- > 1;) pl.push_back((uint64_t *)(base + i * bytes));
            ctx->pool_bytes += (pieces - 1) * bytes;
            *out = (uint64_t *)base;
            return 0;
        }
        (void)hipGetLastError();
    }
    if (hipMalloc((void **)out, bytes) != hipSuccess) {          // out of memory: give the cached arrays back and retry once
        (void)hipGetLastError();
        pool_flush(ctx);
        HIPCHK(hipMalloc((void **)out, bytes));
    }
    return 0;
}
static int dev_release(cn_ctx *ctx, uint64_t *p, size_t bytes) {
    if (ctx->pool_bytes + bytes <= ctx->pool_max || ctx->capturing || in_slab(ctx, p)) { ctx->pool[bytes].push_back(p); ctx->pool_bytes += bytes; return 0; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(p));
    return 0;
}
static int alloc_buf(cn_ctx *ctx, int kind, uint32_t count, uint32_t size, cn_handle *out) {
    if (!out || !count) return fail(CN_ERR_ARG, "bad allocation request");
    Buffer b; b.kind = kind; b.count = count; b.size = size;
    b.item_words = kind == 0 ? (size_t)size * ctx->hc.k * ctx->hc.n : ctx->hc.n;
    CHECK(dev_alloc(ctx, b.item_words * 8 * count, &b.d));
    if (kind == 1) b.pt_zero.assign(count, 1);
    *out = ctx->bufs.insert(std::move(b));
    return 0;
}
// ---- lock-free submission ("defer" = 2): producer side.  A deferrable entry point builds a record and publishes it; the records are executed by ring_drain
// (behind the deferred queue, further down).  submit_async: is the context in that mode?  (read without the lock: the mode changes only through
// cn_set_option, which drains the ring first; a call that races with the change is executed in its claim order either way)
static bool submit_async(const cn_ctx *ctx) { return ctx->defer.load(std::memory_order_relaxed) == 2 && !ctx->capturing.load(std::memory_order_relaxed); }
static int ring_push(cn_ctx *ctx, uint32_t type, uint32_t count, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t x, uint64_t arg);
static void ready_refill(cn_ctx *ctx);
extern "C" int cn_ct_alloc(cn_ctx *ctx, uint32_t count, uint32_t size, cn_handle *out) {
    if (out && count == 1 && size == 2 && submit_async(ctx)) {          // AllocateCiphertext of a per-ciphertext caller: a ready handle, no lock
        const cn_handle h = ctx->ready->pop();
        if (h) { *out = h; return 0; }
    }
    API_BODY
    LOCK_ONLY; if (size < 2 || size > 3) return fail(CN_ERR_ARG, "ciphertext size must be 2 or 3");
    CHECK(alloc_buf(ctx, 0, count, size, out));
    if (count == 1 && size == 2 && ctx->defer.load(std::memory_order_relaxed) == 2) ready_refill(ctx);      // the ring had run dry: fill it while the lock is held anyway
    return 0;
API_END }
extern "C" int cn_pt_alloc(cn_ctx *ctx, uint32_t count, cn_handle *out) { API_BODY LOCK_ONLY; return alloc_buf(ctx, 1, count, 1, out); API_END }
static int free_body(cn_ctx *ctx, cn_handle h);
extern "C" int cn_free(cn_ctx *ctx, cn_handle h) {
    if (submit_async(ctx)) return ring_push(ctx, SUB_FREE, 1, h, 0, 0, 0, 0, 0, 0, 0);      // (a release must not overtake the published calls that read the array)
    API_BODY LOCK_ONLY; return free_body(ctx, h); API_END
}
static int free_body(cn_ctx *ctx, cn_handle h) {
    Buffer *it = ctx->bufs.find(h);
    if (!it) return fail(CN_ERR_ARG, "invalid handle");
    if (it->kind == 2) { NOT_CAPTURING("releasing a GEMM plan"); CHECK(cn_defer_flush(ctx)); CHECK(free_gemm_plan(ctx, *it)); }
    else if (it->kind == 3) { NOT_CAPTURING("releasing a graph"); CHECK(cn_defer_flush(ctx)); CHECK(free_graph(ctx, *it)); }
    else if (cn_defer_pending(ctx)) ctx->dq->frees.emplace_back(it->d, it->item_words * 8 * it->count);   // queued calls may still read it
    else CHECK(dev_release(ctx, it->d, it->item_words * 8 * it->count));
    ctx->bufs.erase(h);
    return 0;
}
// n handles in one call (ReleaseTemp of the unchanged PoolLayer: one Dispose per zero encryption, PoolLayer.cs:83-90; BaseLayer.GetNext: one per column of a
// layer's input, BaseLayer.cs:23-49): the handles are checked first - nothing is released when one of them is invalid - then released like n cn_free calls
static int free_many_body(cn_ctx *ctx, const cn_handle *h, uint32_t n);
extern "C" int cn_free_many(cn_ctx *ctx, const cn_handle *h, uint32_t n) {
    if (submit_async(ctx) && h && n) {
        cn_handle *blk = (cn_handle *)malloc((size_t)n * sizeof(cn_handle));
        if (!blk) return fail(CN_ERR_ARG, "out of host memory");
        memcpy(blk, h, (size_t)n * sizeof(cn_handle));
        return ring_push(ctx, SUB_FREE_MANY, 1, 0, 0, 0, 0, 0, 0, n, (uint64_t)(uintptr_t)blk);
    }
    API_BODY LOCK_ONLY; return free_many_body(ctx, h, n); API_END
}
static int free_many_body(cn_ctx *ctx, const cn_handle *h, uint32_t n) {
    if (!h && n) return fail(CN_ERR_ARG, "null argument");
    bool heavy = false;
    for (uint32_t i = 0; i < n; i++) {
        Buffer *it = ctx->bufs.find(h[i]);
        if (!it) return fail(CN_ERR_ARG, "invalid handle at position %u", i);
        heavy = heavy || it->kind >= 2;
    }
    {   // a handle listed twice would be released twice: refused (sorted copy: the list of a layer's temporaries can be thousands long)
        std::vector<cn_handle> sorted(h, h + n);
        std::sort(sorted.begin(), sorted.end());
        const auto dup = std::adjacent_find(sorted.begin(), sorted.end());
        if (dup != sorted.end()) return fail(CN_ERR_ARG, "handle 0x%llx is listed twice", (unsigned long long)*dup);
    }
    if (heavy) { NOT_CAPTURING("releasing a GEMM plan / a graph"); CHECK(cn_defer_flush(ctx)); }
    const bool pending = cn_defer_pending(ctx);
    for (uint32_t i = 0; i < n; i++) {
        Buffer *it = ctx->bufs.find(h[i]);
        if (it->kind == 2) CHECK(free_gemm_plan(ctx, *it));
        else if (it->kind == 3) CHECK(free_graph(ctx, *it));
        else if (pending) ctx->dq->frees.emplace_back(it->d, it->item_words * 8 * it->count);
        else CHECK(dev_release(ctx, it->d, it->item_words * 8 * it->count));
        ctx->bufs.erase(h[i]);
    }
    return 0;
}
// ---- captured sequences: the launch-bound chains of small kernels of a single-image inference (LoLa: ~235 launches per plaintext
// prime) are recorded once on the context stream and replayed with one hipGraphLaunch - no per-launch host work, dependent kernels
// back to back on the device.  Recording rules: the same sequence must have run once before (so that every temporary comes out of
// the handle pool and the scratch arenas have their size), nothing may synchronise (cn_sync, uploads / downloads of handles, key
// changes) between begin and end, and the handles created while recording must stay alive as long as the graph is launched - the
// kernels carry their addresses.  New inputs go INTO the handles the recorded sequence read (cn_copy / cn_encrypt on them).
static int free_graph(cn_ctx *ctx, Buffer &b) {
    if (!b.cg) return 0;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (b.cg->exec) (void)hipGraphExecDestroy(b.cg->exec);
    if (b.cg->graph) (void)hipGraphDestroy(b.cg->graph);
    for (auto &r : b.cg->reserved) { ctx->pool[r.second].push_back(r.first); ctx->pool_bytes += r.second; }
    b.cg.reset();
    ctx->graphs_alive--;
    return 0;
}
extern "C" int cn_graph_begin(cn_ctx *ctx) { API_BODY
    LOCK; NOT_CAPTURING("cn_graph_begin");
    ctx->cap_staged.clear(); ctx->cap_allocs.clear();
    HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    ctx->capturing = true;
    return 0;
API_END }
extern "C" int cn_graph_end(cn_ctx *ctx, cn_handle *graph) { API_BODY
    LOCK;
    if (!ctx->capturing) return fail(CN_ERR_ARG, "cn_graph_end without cn_graph_begin");
    ctx->capturing = false;
    std::shared_ptr<CapturedGraph> g = std::make_shared<CapturedGraph>();
    hipError_t e = hipStreamEndCapture(ctx->stream, &g->graph);
    if (e != hipSuccess || !g->graph) { (void)hipGetLastError(); ctx->cap_staged.clear(); ctx->cap_allocs.clear(); return fail(CN_ERR_HIP, "graph capture failed: %s", hipGetErrorString(e)); }
    if (!graph) {                                              // nowhere to put the handle: drop the recording (nothing was reserved yet)
        (void)hipGraphDestroy(g->graph); ctx->cap_staged.clear(); ctx->cap_allocs.clear();
        return fail(CN_ERR_ARG, "null argument");
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipGraphDestroy(g->graph); ctx->cap_staged.clear(); ctx->cap_allocs.clear(); return fail(CN_ERR_HIP, "graph instantiation failed: %s", hipGetErrorString(e)); }
    g->staged = std::move(ctx->cap_staged); ctx->cap_staged.clear();
    // arrays handed out while recording that are back in the pool now (temporaries): reserve them for the graph
    for (auto &a : ctx->cap_allocs) {
        auto it = ctx->pool.find(a.second);
        if (it == ctx->pool.end()) continue;
        auto pos = std::find(it->second.begin(), it->second.end(), a.first);
        if (pos == it->second.end()) continue;                  // owned by a live handle (an output of the sequence)
        it->second.erase(pos); ctx->pool_bytes -= a.second;
        g->reserved.push_back(a);
    }
    ctx->cap_allocs.clear();
    Buffer b; b.kind = 3; b.count = 0; b.size = 0; b.d = nullptr; b.item_words = 0; b.cg = g;
    ctx->graphs_alive++;
    *graph = ctx->bufs.insert(std::move(b));
    return 0;
API_END }
extern "C" int cn_graph_launch(cn_ctx *ctx, cn_handle graph) { API_BODY
    LOCK; NOT_CAPTURING("cn_graph_launch");
    Buffer *b = getbuf(ctx, graph, 3);
    if (!b || !b->cg) return fail(CN_ERR_ARG, "invalid graph handle");
    HIPCHK(hipGraphLaunch(b->cg->exec, ctx->stream));
    ctx->st.kernel_launches += 1;
    return 0;
API_END }
extern "C" int cn_live_handles(cn_ctx *ctx) { CnGuard lk(ctx->mu); (void)ring_sync(ctx, false); return (int)ctx->bufs.size() - (int)ctx->ready->size(); }   // (ready handles belong to nobody yet)
extern "C" int cn_ct_upload(cn_ctx *ctx, cn_handle h, uint32_t first, uint32_t count, const uint64_t *host) { API_BODY
    LOCK; NOT_CAPTURING("cn_ct_upload"); GETCT(b, h, 0);
    if (!range_ok(b, first, count)) return fail(CN_ERR_ARG, "index out of range");
    HIPCHK(hipMemcpyAsync(b->d + first * b->item_words, host, count * b->item_words * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
API_END }
extern "C" int cn_ct_download(cn_ctx *ctx, cn_handle h, uint32_t first, uint32_t count, uint64_t *host) { API_BODY
    LOCK; NOT_CAPTURING("cn_ct_download"); GETCT(b, h, 0);
    if (!range_ok(b, first, count)) return fail(CN_ERR_ARG, "index out of range");
    HIPCHK(hipMemcpyAsync(host, b->d + first * b->item_words, count * b->item_words * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
API_END }
extern "C" int cn_pt_upload(cn_ctx *ctx, cn_handle h, uint32_t first, uint32_t count, const uint64_t *host) { API_BODY
    LOCK; NOT_CAPTURING("cn_pt_upload"); GETPT(b, h);
    if (!range_ok(b, first, count)) return fail(CN_ERR_ARG, "index out of range");
    const uint32_t n = ctx->hc.n; const uint64_t t = ctx->hc.t.q;
    for (uint32_t p = 0; p < count; p++) {
        uint8_t z = 1;
        for (uint32_t i = 0; i < n; i++) { uint64_t v = host[(size_t)p * n + i]; if (v >= t) return fail(CN_ERR_ARG, "plaintext coefficient >= plain modulus"); if (v) z = 0; }
        b->pt_zero[first + p] = z;
    }
    HIPCHK(hipMemcpyAsync(b->d + (size_t)first * n, host, (size_t)count * n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
API_END }
extern "C" int cn_pt_download(cn_ctx *ctx, cn_handle h, uint32_t first, uint32_t count, uint64_t *host) { API_BODY
    LOCK; NOT_CAPTURING("cn_pt_download"); GETPT(b, h);
    if (!range_ok(b, first, count)) return fail(CN_ERR_ARG, "index out of range");
    HIPCHK(hipMemcpyAsync(host, b->d + (size_t)first * ctx->hc.n, (size_t)count * ctx->hc.n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
API_END }
// BatchEncoder.Encode / Decode of `count` plaintexts with ONE upload, one scatter launch and one batched (I)NTT mod t: the slot order is
// SEAL's index map (matrix rows -> bit-reversed coefficient positions), kept on the device
__global__ void k_encode_scatter(const uint64_t *__restrict__ values, uint32_t nvalues, const uint32_t *__restrict__ index_map, uint64_t *__restrict__ out, uint32_t n) {
    const uint32_t pt = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[(size_t)pt * n + index_map[i]] = i < nvalues ? values[(size_t)pt * nvalues + i] : 0;
}
__global__ void k_decode_gather(const uint64_t *__restrict__ coeffs, const uint32_t *__restrict__ index_map, uint64_t *__restrict__ values, uint32_t n) {
    const uint32_t pt = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) values[(size_t)pt * n + i] = coeffs[(size_t)pt * n + index_map[i]];
}
static int ensure_index_map(cn_ctx *ctx) {
    if (ctx->d_index_map) return 0;
    HIPCHK(hipMalloc((void **)&ctx->d_index_map, (size_t)ctx->hc.n * 4));
    HIPCHK(hipMemcpy(ctx->d_index_map, ctx->index_map.data(), (size_t)ctx->hc.n * 4, hipMemcpyHostToDevice));
    return 0;
}
// BatchEncoder.Encode: values [count][nvalues] (slot order, each < t; slots beyond nvalues are zero) -> plaintexts pt[pi .. pi + count)
extern "C" int cn_encode_batch(cn_ctx *ctx, const uint64_t *values, uint32_t nvalues, uint32_t count, cn_handle pt, uint32_t pi) { API_BODY
    LOCK; NOT_CAPTURING("cn_encode"); GETPT(b, pt);
    if (!ctx->hc.batching) return fail(CN_ERR_ARG, "plain modulus does not support batching");
    const uint32_t n = ctx->hc.n;
    if (!range_ok(b, pi, count) || nvalues > n || (nvalues && !values)) return fail(CN_ERR_ARG, "bad encode arguments");
    if (!count) return 0;
    const uint64_t t = ctx->hc.t.q;
    std::vector<uint8_t> zero(count, 1);
    for (uint32_t c = 0; c < count; c++) {
        const uint64_t *v = values + (size_t)c * nvalues;
        uint64_t any = 0, big = 0;
        for (uint32_t i = 0; i < nvalues; i++) { any |= v[i]; big |= (uint64_t)(v[i] >= t); }
        if (big) return fail(CN_ERR_ARG, "value >= plain modulus");
        zero[c] = any == 0;
    }
    CHECK(ensure_index_map(ctx));
    uint64_t *d = b->d + (size_t)pi * n;
    if (nvalues) {
        const size_t words = (size_t)count * nvalues;
        CHECK(ensure_scratch(ctx, al(words * 8)));
        uint64_t *stage = salloc<uint64_t>(ctx, words);
        HIPCHK(hipMemcpyAsync(stage, values, words * 8, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_encode_scatter, dim3((n + 255) / 256, count), dim3(256), 0, ctx->stream, stage, nvalues, ctx->d_index_map, d, n);
        HIPCHK(hipGetLastError()); launch_count(ctx);
        HIPCHK(hipStreamSynchronize(ctx->stream));              // the caller's buffer may be released when the call returns
    } else HIPCHK(hipMemsetAsync(d, 0, (size_t)count * n * 8, ctx->stream));
    CHECK(cn_run_ntt(ctx, d, count, ctx->hc.k + ctx->hc.kb, 1, 1));
    for (uint32_t c = 0; c < count; c++) b->pt_zero[pi + c] = zero[c];
    return 0;
API_END }
extern "C" int cn_encode(cn_ctx *ctx, const uint64_t *values, uint32_t nvalues, cn_handle pt, uint32_t pi) { return cn_encode_batch(ctx, values, nvalues, 1, pt, pi); }
// BatchEncoder.Decode: plaintexts pt[pi .. pi + count) -> values [count][N] in slot order
extern "C" int cn_decode_batch(cn_ctx *ctx, cn_handle pt, uint32_t pi, uint32_t count, uint64_t *values) { API_BODY
    LOCK; NOT_CAPTURING("cn_decode"); GETPT(b, pt);
    if (!ctx->hc.batching) return fail(CN_ERR_ARG, "plain modulus does not support batching");
    const uint32_t n = ctx->hc.n;
    if (!range_ok(b, pi, count) || !values) return fail(CN_ERR_ARG, "bad decode arguments");
    if (!count) return 0;
    CHECK(ensure_index_map(ctx));
    const size_t words = (size_t)count * n;
    CHECK(ensure_scratch(ctx, 2 * al(words * 8)));
    uint64_t *tmp = salloc<uint64_t>(ctx, words), *slots = salloc<uint64_t>(ctx, words);
    HIPCHK(hipMemcpyAsync(tmp, b->d + (size_t)pi * n, words * 8, hipMemcpyDeviceToDevice, ctx->stream));
    CHECK(cn_run_ntt(ctx, tmp, count, ctx->hc.k + ctx->hc.kb, 1, 0));
    hipLaunchKernelGGL(k_decode_gather, dim3((n + 255) / 256, count), dim3(256), 0, ctx->stream, tmp, ctx->d_index_map, slots, n);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    HIPCHK(hipMemcpyAsync(values, slots, words * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
API_END }
extern "C" int cn_decode(cn_ctx *ctx, cn_handle pt, uint32_t pi, uint64_t *values) { return cn_decode_batch(ctx, pt, pi, 1, values); }
extern "C" int cn_copy(cn_ctx *ctx, cn_handle src, uint32_t sfirst, cn_handle dst, uint32_t dfirst, uint32_t count) { API_BODY
    LOCK_ONLY;
    Buffer *s = ctx->bufs.find(src), *d = ctx->bufs.find(dst);
    if (deferring(ctx) && s && d && s->kind == 0 && d->kind == 0 && s->size == 2 && d->size == 2 && count && count <= DEFER_STAGED_MAX &&
        range_ok(s, sfirst, count) && range_ok(d, dfirst, count) && !(s == d && sfirst < dfirst + count && dfirst < sfirst + count))
        return defer_staged(ctx, DOP_COPY, s, sfirst, nullptr, 0, nullptr, 0, d, dfirst, count, 0);
    CHECK(cn_defer_flush(ctx));
    if (!s || !d) return fail(CN_ERR_ARG, "invalid handle");
    if (s->kind != d->kind || s->item_words != d->item_words) return fail(CN_ERR_ARG, "copy between different buffer shapes");
    if (!range_ok(s, sfirst, count) || !range_ok(d, dfirst, count)) return fail(CN_ERR_ARG, "index out of range");
    HIPCHK(hipMemcpyAsync(d->d + dfirst * d->item_words, s->d + sfirst * s->item_words, count * s->item_words * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (s->kind == 1) for (uint32_t i = 0; i < count; i++) d->pt_zero[dfirst + i] = s->pt_zero[sfirst + i];
    return 0;
API_END }
extern "C" int cn_device_ptr(cn_ctx *ctx, cn_handle h, void **ptr, size_t *bytes) {
    CnGuard lk(ctx->mu);
    Buffer *it = ctx->bufs.find(h);
    if (!it) return fail(CN_ERR_ARG, "invalid handle");
    if (ptr) *ptr = it->d;
    if (bytes) *bytes = it->item_words * 8 * it->count;
    return 0;
}

// ---------------------------------------------------------------- linear ops
static int addsub(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count, int op) {
    GETCT(A, a, 0); GETCT(O, out, A->size);
    Buffer *B = A;
    if (op != 2) { B = getbuf(ctx, b, 0); if (!B || B->size != A->size) return fail(CN_ERR_ARG, "operand sizes do not match"); }
    if (!range_ok(A, ai, count) || !range_ok(O, oi, count) || (op != 2 && !range_ok(B, bi, count))) return fail(CN_ERR_ARG, "index out of range");
    if (!count) return 0;
    uint32_t limbs = count * A->size * ctx->hc.k;
    hipLaunchKernelGGL(k_addsub, dim3(limbs * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, A->d + ai * A->item_words,
                       B->d + (op != 2 ? bi : ai) * B->item_words, O->d + oi * O->item_words, ctx->dc, ctx->chunks, op);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    return 0;
}
static bool deferring(cn_ctx *ctx);
static int defer_addsub(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count, int op);
static int defer_add_plain(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, int subtract, cn_handle out, uint32_t oi, uint32_t count);
static int defer_mul_relin(cn_ctx *ctx, cn_handle a, uint32_t ai, uint32_t astride, cn_handle b, uint32_t bi, uint32_t bstride, cn_handle out, uint32_t oi, uint32_t count);
static int addsub_body(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count, int op) {
    if (deferring(ctx)) { int rc = defer_addsub(ctx, a, ai, b, bi, out, oi, count, op); if (rc <= 0) return rc; }      // > 0: not deferrable (size-3 operands)
    CHECK(cn_defer_flush(ctx)); CHECK(addsub(ctx, a, ai, b, bi, out, oi, count, op));
    if (op) ctx->st.Subtraction += count; else ctx->st.Addition += count;
    return 0;
}
extern "C" int cn_add(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count) {
    if (submit_async(ctx) && count <= DEFER_STAGED_MAX) return ring_push(ctx, SUB_ADD, count, a, ai, b, bi, out, oi, 0, 0);
    API_BODY LOCK_ONLY; return addsub_body(ctx, a, ai, b, bi, out, oi, count, 0); API_END
}
extern "C" int cn_sub(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count) {
    if (submit_async(ctx) && count <= DEFER_STAGED_MAX) return ring_push(ctx, SUB_SUB, count, a, ai, b, bi, out, oi, 0, 0);
    API_BODY LOCK_ONLY; return addsub_body(ctx, a, ai, b, bi, out, oi, count, 1); API_END
}
extern "C" int cn_negate(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle out, uint32_t oi, uint32_t count) { API_BODY
    LOCK; return addsub(ctx, a, ai, a, ai, out, oi, count, 2);
API_END }
extern "C" int cn_add_many(cn_ctx *ctx, cn_handle in, const uint32_t *idx, uint32_t n_idx, cn_handle out, uint32_t oi) { API_BODY
    LOCK; GETCT(I, in, 0); GETCT(O, out, I->size);
    if (!n_idx || !idx) return fail(CN_ERR_ARG, "AddMany of an empty list");
    for (uint32_t i = 0; i < n_idx; i++) if (idx[i] >= I->count) return fail(CN_ERR_ARG, "index out of range");
    if (oi >= O->count) return fail(CN_ERR_ARG, "index out of range");
    CHECK(ensure_scratch(ctx, al(n_idx * 4)));
    uint32_t *didx; CHECK(upload_tmp(ctx, idx, n_idx, &didx));
    uint32_t limbs = I->size * ctx->hc.k;
    hipLaunchKernelGGL(k_add_many, dim3(limbs * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, I->d, didx, n_idx, I->item_words,
                       O->d + oi * O->item_words, ctx->dc, ctx->chunks);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    ctx->st.AddMany += 1; ctx->st.AddManyItemCount += n_idx;
    return 0;
API_END }
static int add_plain_body(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, int subtract, cn_handle out, uint32_t oi, uint32_t count);
extern "C" int cn_add_plain(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, int subtract, cn_handle out, uint32_t oi, uint32_t count) {
    if (submit_async(ctx) && count <= DEFER_STAGED_MAX) return ring_push(ctx, SUB_ADD_PLAIN, count, a, ai, pt, pi, out, oi, subtract ? 1u : 0u, 0);
    API_BODY LOCK_ONLY; return add_plain_body(ctx, a, ai, pt, pi, subtract, out, oi, count); API_END
}
static int add_plain_body(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, int subtract, cn_handle out, uint32_t oi, uint32_t count) {
    if (deferring(ctx)) { int rc = defer_add_plain(ctx, a, ai, pt, pi, subtract, out, oi, count); if (rc <= 0) return rc; }
    CHECK(cn_defer_flush(ctx));
    GETCT(A, a, 0); GETCT(O, out, A->size); GETPT(P, pt);
    if (!range_ok(A, ai, count) || !range_ok(O, oi, count) || !range_ok(P, pi, count)) return fail(CN_ERR_ARG, "index out of range");
    if (!count) return 0;
    uint32_t limbs = count * A->size * ctx->hc.k;
    hipLaunchKernelGGL(k_add_plain, dim3(limbs * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, A->d + ai * A->item_words,
                       P->d + (size_t)pi * ctx->hc.n, ctx->hc.n, O->d + oi * O->item_words, ctx->dc, ctx->chunks, A->size, subtract);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    if (subtract) ctx->st.PlainSubtraction += count; else ctx->st.PlainAddition += count;
    return 0;
}
// out[c] = a[c * (a_bcast ? 0 : 1)] * pt[c * pstride]; a_bcast: ONE ciphertext against `count` plaintexts (row-dot batches)
// Dense MultiplyPlain in two launches (k_lift_ntt, k_mul_plain_fused); ranges / zero plaintexts were checked by the caller
// A row-dot batch whose SumAllSlots chain follows: the product kernel leaves sigma_elt(c1) of every product in `out` ([row][k][N], the chain's first scratch array) and takes
// its transformed ciphertext from `ctn` - both inside the scratch arena the caller has sized for the whole call (no ensure_scratch in between: the arena must not move)
struct BcastNext { uint64_t elt; uint64_t *out, *ctn; };
static int mul_plain_fused(cn_ctx *ctx, Buffer *A, uint32_t ai, bool a_bcast, Buffer *P, uint32_t pi, uint32_t pstride, Buffer *O, uint32_t oi, uint32_t count,
                           const BcastNext *nx = nullptr) {
    const uint32_t n = ctx->hc.n, k = ctx->hc.k, npt = pstride ? count : 1;
    uint64_t *o = O->d + oi * O->item_words;
    const uint64_t *src = A->d + ai * A->item_words;
    // one input ciphertext broadcast over the outputs: it must survive until the last block has read it
    bool f64 = ctx->use_f64, light = true;
    for (uint32_t m = 0; m < k; m++) { f64 = f64 && ctx->hc.f64ok[m]; if (ctx->hc.q[m].q >> 44) light = false; }
    const int pol = f64 && light ? POL_F64L : (f64 ? POL_F64 : POL_U64);
    if (a_bcast && pstride && count >= 4 && ctx->mp_bcast) {      // one ciphertext x many plaintexts: transform the ciphertext once, the plaintexts inside the product kernel
        uint64_t *ctn = nx ? nx->ctn : nullptr;
        if (!nx) { CHECK(ensure_scratch(ctx, al(A->item_words * 8))); ctn = salloc<uint64_t>(ctx, A->item_words); }
        if (!ctn) return fail(CN_ERR_HIP, "internal: scratch exhausted in multiply_plain");
        HIPCHK(hipMemcpyAsync(ctn, src, A->item_words * 8, hipMemcpyDeviceToDevice, ctx->stream));
        CHECK(cn_run_ntt(ctx, ctn, A->size * k, 0, k, 0));
        rr_ops[pol]->mul_plain_bcast(ctx, P->d + (size_t)pi * n, pstride, ctn, o, count, A->size, nx ? (uint32_t)nx->elt : 0u, nx ? nx->out : nullptr);
        HIPCHK(hipGetLastError()); launch_count(ctx);
        ctx->st.ntt_forward_limbs += (uint64_t)A->size * k + (uint64_t)count * A->size * k; ctx->st.ntt_inverse_limbs += (uint64_t)count * A->size * k;
        ctx->st.PlainMultiplication += count;
        return 0;
    }
    const bool alias = a_bcast && src >= o && src < o + (size_t)count * A->item_words;
    CHECK(ensure_scratch(ctx, al((size_t)npt * k * n * 8) + (alias ? al(A->item_words * 8) : 0)));
    uint64_t *lift = salloc<uint64_t>(ctx, (size_t)npt * k * n);
    if (alias) {
        uint64_t *keep = salloc<uint64_t>(ctx, A->item_words);
        if (!keep) return fail(CN_ERR_HIP, "internal: scratch exhausted in multiply_plain");
        HIPCHK(hipMemcpyAsync(keep, src, A->item_words * 8, hipMemcpyDeviceToDevice, ctx->stream));
        src = keep;
    }
    if (!lift) return fail(CN_ERR_HIP, "internal: scratch exhausted in multiply_plain");
    const uint64_t *pt = P->d + (size_t)pi * n;
    const size_t sstride = a_bcast ? 0 : A->item_words;
    const uint32_t pitch = pstride ? pstride : 1u, ps = pstride ? 1u : 0u;
    rr_ops[pol]->mul_plain_fused(ctx, pt, pitch, npt, lift, src, sstride, ps, o, count, A->size);
    HIPCHK(hipGetLastError()); launch_count(ctx, 2);
    ctx->st.ntt_forward_limbs += (uint64_t)npt * k + (uint64_t)count * A->size * k; ctx->st.ntt_inverse_limbs += (uint64_t)count * A->size * k;
    ctx->st.PlainMultiplication += count;
    return 0;
}
static bool mul_plain_takes_bcast(cn_ctx *ctx, uint32_t count) { return ctx->mp_fused && ctx->mp_bcast && !ctx->legacy_ntt && ctx->hc.logn >= 10 && ctx->hc.logn <= 14 && count >= 4; }
static int mul_plain_impl(cn_ctx *ctx, Buffer *A, uint32_t ai, bool a_bcast, Buffer *P, uint32_t pi, uint32_t pstride, Buffer *O, uint32_t oi, uint32_t count,
                          const BcastNext *nx = nullptr) {
    if (!range_ok(A, ai, a_bcast ? 1 : count) || !range_ok(O, oi, count) || !range_ok(P, pi, pstride ? count : 1, pstride ? pstride : 1))
        return fail(CN_ERR_ARG, "index out of range");
    if (!count) return 0;
    if (A == O && !a_bcast && ai != oi && ai < oi + count && oi < ai + count)          // block c writes out[c] while another block still reads in[c']
        return fail(CN_ERR_ARG, "multiply_plain: input and output ranges overlap partially (use the same range or disjoint ranges)");
    const uint32_t n = ctx->hc.n, k = ctx->hc.k, npt = pstride ? count : 1;
    for (uint32_t c = 0; c < npt; c++) if (P->pt_zero[pi + c * pstride]) return fail(CN_ERR_ZERO, "plain cannot be zero");
    if (ctx->mp_fused && !ctx->legacy_ntt && ctx->hc.logn >= 10 && ctx->hc.logn <= 14) return mul_plain_fused(ctx, A, ai, a_bcast, P, pi, pstride, O, oi, count, nx);
    if (nx) return fail(CN_ERR_ARG, "internal: chained row-dot batch outside the fused product");
    CHECK(ensure_scratch(ctx, al((size_t)npt * k * n * 8)));
    uint64_t *lift = salloc<uint64_t>(ctx, (size_t)npt * k * n);
    // lift every referenced plaintext into the k limbs (one launch), NTT them
    hipLaunchKernelGGL(k_lift_plain, dim3(npt * k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, P->d + (size_t)pi * n, lift, ctx->dc, ctx->chunks, pstride ? pstride : 1u);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    CHECK(cn_run_ntt(ctx, lift, npt * k, 0, k, 0));
    uint64_t *o = O->d + oi * O->item_words;
    const uint64_t *src = A->d + ai * A->item_words;
    if (a_bcast) {
        for (uint32_t c = 0; c < count; c++)
            if (o + c * A->item_words != src) HIPCHK(hipMemcpyAsync(o + c * A->item_words, src, A->item_words * 8, hipMemcpyDeviceToDevice, ctx->stream));
    } else if (o != src) HIPCHK(hipMemcpyAsync(o, src, count * A->item_words * 8, hipMemcpyDeviceToDevice, ctx->stream));
    uint32_t limbs = count * A->size * k;
    CHECK(cn_run_ntt(ctx, o, limbs, 0, k, 0));
    hipLaunchKernelGGL(k_dyadic_pt, dim3(limbs * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, o, lift, pstride ? 1u : 0u, ctx->dc, ctx->chunks, A->size);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    CHECK(cn_run_ntt(ctx, o, limbs, 0, k, 1));
    ctx->st.PlainMultiplication += count;
    return 0;
}
extern "C" int cn_mul_plain(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, uint32_t pstride, cn_handle out, uint32_t oi, uint32_t count) { API_BODY
    LOCK_ONLY; GETCT(A, a, 0); GETCT(O, out, A->size); GETPT(P, pt);
    if (deferring(ctx) && count && count <= DEFER_STAGED_MAX && A->size == 2) {       // the per-row MultiplyPlain of an unchanged caller: queued, rows merged at flush
        if (!range_ok(A, ai, count) || !range_ok(O, oi, count) || !range_ok(P, pi, pstride ? count : 1, pstride ? pstride : 1)) return fail(CN_ERR_ARG, "index out of range");
        for (uint32_t c = 0; c < (pstride ? count : 1u); c++) if (P->pt_zero[pi + c * pstride]) return fail(CN_ERR_ZERO, "plain cannot be zero");
        if (A == O && ai != oi && ai < oi + count && oi < ai + count) return fail(CN_ERR_ARG, "multiply_plain: input and output ranges overlap partially (use the same range or disjoint ranges)");
        return defer_staged(ctx, DOP_MULPLAIN, A, ai, nullptr, 0, P->d + (size_t)pi * ctx->hc.n, pstride * ctx->hc.n, O, oi, count, 0);
    }
    CHECK(cn_defer_flush(ctx));
    return mul_plain_impl(ctx, A, ai, false, P, pi, pstride, O, oi, count);
API_END }
static uint64_t lift_scalar(const DevConsts &hc, uint64_t w, uint32_t j) { return w >= hc.t_half ? w + hc.lift_inc[j] : w; }
extern "C" int cn_mul_scalar(cn_ctx *ctx, cn_handle a, uint32_t ai, const uint64_t *scalars, uint32_t sstride, cn_handle out, uint32_t oi, uint32_t count) { API_BODY
    LOCK; GETCT(A, a, 0); GETCT(O, out, A->size);
    if (!range_ok(A, ai, count) || !range_ok(O, oi, count) || !scalars) return fail(CN_ERR_ARG, "index out of range");
    if (!count) return 0;
    const uint32_t k = ctx->hc.k, ns = sstride ? count : 1;
    std::vector<uint64_t> sc((size_t)ns * k);
    for (uint32_t c = 0; c < ns; c++) {
        uint64_t w = scalars[(size_t)c * sstride];
        if (w >= ctx->hc.t.q) return fail(CN_ERR_ARG, "scalar >= plain modulus");
        if (!w) return fail(CN_ERR_ZERO, "plain cannot be zero");
        for (uint32_t j = 0; j < k; j++) sc[(size_t)c * k + j] = lift_scalar(ctx->hc, w, j);
    }
    CHECK(ensure_scratch(ctx, al(sc.size() * 8)));
    uint64_t *dsc; CHECK(upload_tmp(ctx, sc.data(), sc.size(), &dsc));
    uint32_t limbs = count * A->size * k;
    hipLaunchKernelGGL(k_mul_scalar, dim3(limbs * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, A->d + ai * A->item_words, dsc, sstride ? 1u : 0u,
                       O->d + oi * O->item_words, ctx->dc, ctx->chunks, A->size);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    ctx->st.PlainMultiplication += count;
    return 0;
API_END }

// HOT LOOP A
// weight tiles of a planned GEMM in kernel layout; row(g, m): the K weights (residues mod t) of member m of group g, or null
// tap_ok(g, kk): term kk of group g gathers a ciphertext; a padded tap gets the weight 0 whatever the caller passed (k_scalar_gemm_f64 multiplies a valid word
// by it instead of selecting per lane)
template <class ROW, class TAP> static void pack_gemm_weights(cn_ctx *ctx, uint32_t G, uint32_t M, uint32_t K, bool small, ROW row, TAP tap_ok, uint32_t &MT, std::vector<char> &wbytes) {
    const uint32_t k = ctx->hc.k; const uint64_t t = ctx->hc.t.q;
    if (small) {
        const uint32_t MTf = M >= 16 ? 20 : (M >= 8 ? 10 : (M >= 3 ? 5 : 1)), mtf = (M + MTf - 1) / MTf;
        const uint32_t Kw = gemm_f64_rows(K);
        std::vector<double> hWd((size_t)G * mtf * Kw * MTf, 0.0);           // [g][mtile][kk < Kw][m], zero padded (gemm_f64_rows)
        for (uint32_t g = 0; g < G; g++) for (uint32_t m = 0; m < M; m++) {
            const uint64_t *wr = row(g, m);
            if (!wr) continue;
            double *dst = &hWd[(((size_t)g * mtf + m / MTf) * Kw) * MTf + m % MTf];
            for (uint32_t kk = 0; kk < K; kk++) { uint64_t w = tap_ok(g, kk) ? wr[kk] : 0; dst[(size_t)kk * MTf] = w >= ctx->hc.t_half ? -(double)(t - w) : (double)w; }
        }
        MT = MTf;
        wbytes.assign((const char *)hWd.data(), (const char *)(hWd.data() + hWd.size()));
    } else {
        const uint32_t MTi = M >= 8 ? 10 : (M >= 3 ? 5 : 1), mti = (M + MTi - 1) / MTi;
        std::vector<uint64_t> hW((size_t)k * G * mti * K * MTi, 0);         // [j][g][mtile][kk][m], zero padded
        for (uint32_t j = 0; j < k; j++) for (uint32_t g = 0; g < G; g++) for (uint32_t m = 0; m < M; m++) {
            const uint64_t *wr = row(g, m);
            if (!wr) continue;
            uint64_t *dst = &hW[((((size_t)j * G + g) * mti + m / MTi) * K) * MTi + m % MTi];
            for (uint32_t kk = 0; kk < K; kk++) { uint64_t w = tap_ok(g, kk) ? wr[kk] : 0; dst[(size_t)kk * MTi] = w ? lift_scalar(ctx->hc, w, j) : 0; }
        }
        MT = MTi;
        wbytes.assign((const char *)hW.data(), (const char *)(hW.data() + hW.size()));
    }
}
// small signed weights (|w| < 2^20 after centring mod t - every PoolLayer weight round(w*scale) is): exact-FP64 limb-split kernel
static bool gemm_weights_small(cn_ctx *ctx, const uint64_t *W, size_t count) {
    for (size_t x = 0; x < count; x++) {
        const uint64_t w = W[x], a = w >= ctx->hc.t_half ? ctx->hc.t.q - w : w;
        if (a >> 20) return false;
    }
    return true;
}
struct GemmArith { bool small, two; uint32_t lazy; int bits; };
static GemmArith gemm_arith(cn_ctx *ctx, bool weights_small) {
    uint64_t qmax = 0; for (uint32_t j = 0; j < ctx->hc.k; j++) qmax = std::max(qmax, ctx->hc.q[j].q);
    const int bits = 64 - __builtin_clzll(qmax);
    GemmArith g;
    g.bits = bits;
    g.small = weights_small && ctx->use_f64 && bits <= 49;       // the kernel folds its limb sums with exact-FP64 modular arithmetic (q < 2^49.4)
    g.two = bits <= 44;                                          // 2 limbs of 22 bits, else 3 limbs of 17 bits
    if (g.small) g.lazy = g.two ? 1024u : 32768u;                // terms whose limb products (< 2^42 / 2^37) still sum exactly below 2^52
    else g.lazy = (2 * bits >= 127) ? 1u : (uint32_t)std::min<uint64_t>(1u << 20, 1ull << (127 - 2 * bits));   // products of two values < q_max in 128 bits
    return g;
}

// One-limb form of the small-weight kernel: sum_k w_k x_k with x_k < q_max is an exact double as long as (sum_k |w_k| + 1) q_max <= 2^53 for every output row
// (all partial sums are integers below 2^53; the + 1 leaves room for the recentred carry of the fold) - the words are then not split into limbs at all: ONE FMA
// per MAC instead of two, no masks and shifts, one recentring per output.  True for the CryptoNets convolution (row sums <= 373 with the trained weights,
// 44-bit moduli); the dense layers have larger row sums and keep the two-limb form (or the matrix cores).  Needs the whole term list in one block (K <= lazy).
template <class ROW, class TAP> static bool gemm_one_limb(cn_ctx *ctx, const GemmArith &ar, uint32_t G, uint32_t M, uint32_t K, ROW row, TAP tap_ok) {
    static const bool on = !(getenv("CN_GEMM_ONE_LIMB") && !atoi(getenv("CN_GEMM_ONE_LIMB")));
    if (!on || !ar.small || ar.bits > 49 || K > ar.lazy) return false;
    uint64_t qmax = 0; for (uint32_t j = 0; j < ctx->hc.k; j++) qmax = std::max(qmax, ctx->hc.q[j].q);
    const uint64_t room = (1ull << 53) / qmax;                 // sum |w| + 1 <= room
    const uint64_t t = ctx->hc.t.q;
    for (uint32_t g = 0; g < G; g++) for (uint32_t m = 0; m < M; m++) {
        const uint64_t *wr = row(g, m);
        if (!wr) continue;
        uint64_t sum = 1;
        for (uint32_t kk = 0; kk < K; kk++) if (tap_ok(g, kk)) { const uint64_t w = wr[kk]; sum += w >= ctx->hc.t_half ? t - w : w; if (sum > room) return false; }
    }
    return true;
}

// ---- the matrix-core form of a scalar GEMM (k_scalar_gemm_mfma): eligibility, weight digit planes, A fragments
// 6 signed base-256 digits cover residues below 2^46 (x + 0x80..80 must stay below 2^48); i32 accumulators hold K * P * 2^14 < 2^31
static bool gemm_mfma_ok(cn_ctx *ctx, const GemmArith &ar, uint32_t M, uint32_t K) {
    return ctx->gemm_mfma && ar.small && ar.bits <= 46 && M >= 16 && (uint64_t)K * 3 < (1u << 17) && !(ctx->hc.n & 31);
}
static uint32_t gemm_weight_planes(cn_ctx *ctx, const uint64_t *W, size_t count) {
    uint64_t amax = 0;
    for (size_t x = 0; x < count; x++) { const uint64_t w = W[x]; amax = std::max(amax, w >= ctx->hc.t_half ? ctx->hc.t.q - w : w); }
    return amax <= 127 ? 1u : (amax <= 32639 ? 2u : 3u);          // signed digits -128..127: |w| <= 127 / 32639 / 8355711
}
// fragments [g][p][mtile][kstep][lane][16]: lane l, byte t = digit p of the weight of output row 32 mtile + (l & 31) for term
// 32 kstep + 16 (l >> 5) + t; zero for padded rows / terms / taps.  row(g, m): residues mod t of member m, or null; tap_ok(g, kk).
template <class ROW, class TAP> static void pack_gemm_mfma(cn_ctx *ctx, uint32_t G, uint32_t M, uint32_t K, uint32_t P, ROW row, TAP tap_ok, std::vector<char> &wbytes) {
    const uint32_t mtiles = (M + 31) / 32, ksteps = (K + 31) / 32;
    const uint64_t t = ctx->hc.t.q;
    wbytes.assign((size_t)G * P * mtiles * ksteps * 1024, 0);
    for (uint32_t g = 0; g < G; g++) for (uint32_t m = 0; m < M; m++) {
        const uint64_t *wr = row(g, m);
        if (!wr) continue;
        const uint32_t mt = m / 32, r = m % 32;
        for (uint32_t kk = 0; kk < K; kk++) {
            const uint64_t w = wr[kk];
            if (!w || !tap_ok(g, kk)) continue;
            const int64_t sw = w >= ctx->hc.t_half ? -(int64_t)(t - w) : (int64_t)w;
            const uint32_t rec = ((uint32_t)(sw + 0x808080) ^ 0x808080u);          // byte p = signed digit p
            const uint32_t ks = kk / 32, half = (kk % 32) / 16, tt = kk % 16;
            for (uint32_t p = 0; p < P; p++)
                wbytes[((((size_t)g * P + p) * mtiles + mt) * ksteps + ks) * 1024 + (size_t)(half * 32 + r) * 16 + tt] = (char)(uint8_t)(rec >> (8 * p));
        }
    }
}
// A scalar GEMM is planned once per (gather table, weight matrix): validation, grouping of the outputs that share a gather list,
// weight tiles in the kernel's layout.  A plan can live in HBM (cn_gemm_plan_create: the weights of a layer are uploaded once, every
// inference only launches) or in the per-call scratch (cn_scalar_gemm).
struct GemmPlan {
    uint32_t O = 0, K = 0, Kp = 0, G = 0, M = 0, MT = 0, lazy = 0, max_in = 0;
    bool small = false, two = false, one = false, has_bias = false, mfma = false;
    uint32_t P = 0, mtiles = 0, ksteps = 0;  // matrix-core form: weight digit planes, 32-row output tiles, 32-term steps
    cn_handle bias_pt = 0; uint32_t bias_count = 0;
    uint64_t nnz = 0;                        // non-zero, non-padded terms (statistics)
    std::vector<char> host;                  // [idx | out_idx | bias_idx | weights], each 256 B aligned
    size_t off_oidx = 0, off_bidx = 0, off_w = 0;
    char *dev = nullptr;                     // persistent plans: device copy of `host`
};
static int free_gemm_plan(cn_ctx *ctx, Buffer &b) {
    if (b.plan && b.plan->dev) { HIPCHK(hipStreamSynchronize(ctx->stream)); HIPCHK(hipFree(b.plan->dev)); b.plan->dev = nullptr; }
    return 0;
}
// Gather lists that overlap are merged in PAIRS (round 5).  A convolution window of 25 taps shares 15 of them with its neighbour; as two gather lists every shared
// input word travels from L2 to a CU twice - and that traffic, not HBM and not instruction issue, is what the layer waits for (profiles/HISTORY.md, round 5: the
// CryptoNets convolution 370-380 us with one list per window, 300-320 us with the windows in pairs; wider tiles lose again: their outputs no longer fit the
// 10-output register tile).  Two lists that share at least half of their inputs become ONE list (the union, 35 entries for two neighbouring 5 x 5 windows at stride
// 2) whose outputs carry the weight 0 for the entries of the other list: a zero weight is "no term" in DenseMatrixBySparseVectorMultiply, the outputs are the same
// words.  Only for small signed weights (the FP64 kernels), lists of at most 64 entries without repeated inputs and at most 5 outputs each (a pair then fills the
// 10-output tile).  gidx / W2 / K are rewritten in place; returns false when nothing was merged.
static bool pair_gather_lists(uint32_t O, uint32_t &K, std::vector<int32_t> &gidx, const uint64_t *W, std::vector<uint64_t> &W2) {
    if (K > 64 || O < 2) return false;
    std::map<std::vector<int32_t>, std::vector<uint32_t>> groups;
    for (uint32_t o = 0; o < O; o++) groups[std::vector<int32_t>(gidx.begin() + (size_t)o * K, gidx.begin() + (size_t)(o + 1) * K)].push_back(o);
    const size_t G = groups.size();
    if (G < 2 || G > 65536) return false;
    struct L { std::vector<int32_t> in; const std::vector<uint32_t> *outs; const std::vector<int32_t> *list; int32_t mate = -1; };
    std::vector<L> ls; ls.reserve(G);
    for (auto &kv : groups) {
        if (kv.second.size() > 5) return false;
        L l; l.outs = &kv.second; l.list = &kv.first;
        for (int32_t id : kv.first) if (id >= 0) l.in.push_back(id);
        std::sort(l.in.begin(), l.in.end());
        if (std::adjacent_find(l.in.begin(), l.in.end()) != l.in.end()) return false;       // an input twice in one list: its two weights would have to be added
        ls.push_back(std::move(l));
    }
    std::unordered_map<int32_t, std::vector<uint32_t>> where;                               // input -> lists that gather it
    for (uint32_t i = 0; i < G; i++) for (int32_t id : ls[i].in) where[id].push_back(i);
    bool any = false;
    std::vector<uint32_t> cnt(G, 0), touched;
    for (uint32_t i = 0; i < G; i++) {
        if (ls[i].mate >= 0) continue;
        touched.clear();
        for (int32_t id : ls[i].in) for (uint32_t j : where[id]) if (j > i && ls[j].mate < 0) { if (!cnt[j]++) touched.push_back(j); }
        uint32_t best = 0; int32_t bj = -1;
        for (uint32_t j : touched) { if (cnt[j] > best || (cnt[j] == best && (int32_t)j < bj)) { best = cnt[j]; bj = (int32_t)j; } cnt[j] = 0; }
        if (bj >= 0 && 2 * best >= std::min(ls[i].in.size(), ls[bj].in.size()) && ls[i].in.size() + ls[bj].in.size() - best <= 64) { ls[i].mate = bj; ls[bj].mate = (int32_t)i; any = true; }
    }
    if (!any) return false;
    // the union of a pair: the first list's entries in their order, then the second list's new ones; K2 = the longest list after merging
    std::vector<std::vector<int32_t>> uni(G);
    uint32_t K2 = 0;
    for (uint32_t i = 0; i < G; i++) {
        const int32_t m = ls[i].mate;
        if (m >= 0 && (uint32_t)m < i) { uni[i] = uni[m]; continue; }
        for (int32_t id : *ls[i].list) if (id >= 0) uni[i].push_back(id);
        if (m >= 0) for (int32_t id : *ls[m].list) if (id >= 0 && !std::binary_search(ls[i].in.begin(), ls[i].in.end(), id)) uni[i].push_back(id);
        K2 = std::max<uint32_t>(K2, (uint32_t)uni[i].size());
    }
    std::vector<int32_t> g2((size_t)O * K2, -1);
    W2.assign((size_t)O * K2, 0);
    for (uint32_t i = 0; i < G; i++) {
        std::unordered_map<int32_t, uint32_t> pos;
        for (uint32_t x = 0; x < uni[i].size(); x++) pos[uni[i][x]] = x;
        for (uint32_t o : *ls[i].outs) {
            for (uint32_t x = 0; x < uni[i].size(); x++) g2[(size_t)o * K2 + x] = uni[i][x];
            for (uint32_t kk = 0; kk < K; kk++) { const int32_t id = gidx[(size_t)o * K + kk]; if (id >= 0) W2[(size_t)o * K2 + pos[id]] = W[(size_t)o * K + kk]; }
        }
    }
    gidx.swap(g2); K = K2;
    return true;
}
static int build_gemm_plan(cn_ctx *ctx, const int32_t *idx, const uint64_t *W, uint32_t O, uint32_t K, Buffer *BP, cn_handle bias_pt, const int32_t *bias_idx,
                           GemmPlan &P) {
    if (!O || !K || !W) return fail(CN_ERR_ARG, "empty scalar GEMM");
    if (bias_pt && (!BP || !bias_idx)) return fail(CN_ERR_ARG, "invalid bias plaintext handle");
    const uint64_t t = ctx->hc.t.q;
    // validate + default gather (identity) + reference semantics: zero weights are skipped, all-zero row is an error
    std::vector<int32_t> gidx((size_t)O * K);
    for (uint32_t o = 0; o < O; o++) {
        bool any = false;
        for (uint32_t kk = 0; kk < K; kk++) {
            int32_t id = idx ? idx[(size_t)o * K + kk] : (int32_t)kk;
            uint64_t w = W[(size_t)o * K + kk];
            if (w >= t) return fail(CN_ERR_ARG, "weight >= plain modulus");
            if (id >= 0) P.max_in = std::max<uint32_t>(P.max_in, (uint32_t)id + 1);
            if (id >= 0 && w) { any = true; P.nnz++; }
            gidx[(size_t)o * K + kk] = id;
        }
        if (!any) return fail(CN_ERR_ARG, "output %u has no non-zero term (AddMany of nothing)", o);
        if (BP && (bias_idx[o] < 0 || (uint32_t)bias_idx[o] >= BP->count)) return fail(CN_ERR_ARG, "bias index out of range");
    }
    std::vector<uint64_t> W2;
    if (ctx->gemm_pair && gemm_arith(ctx, gemm_weights_small(ctx, W, (size_t)O * K)).small && pair_gather_lists(O, K, gidx, W, W2)) W = W2.data();
    // group outputs that gather the same inputs (PoolLayer: every map of one corner shares its patch)
    std::map<std::vector<int32_t>, std::vector<uint32_t>> groups;
    for (uint32_t o = 0; o < O; o++) groups[std::vector<int32_t>(gidx.begin() + (size_t)o * K, gidx.begin() + (size_t)(o + 1) * K)].push_back(o);
    // groups may differ in size (a tiled convolution has smaller tiles at the border): M = the largest, the missing members of
    // smaller groups get output index -1 (nothing stored) and all-zero weight rows
    const uint32_t NONE = 0xffffffffu;
    uint32_t G = (uint32_t)groups.size(), M = 0;
    for (auto &g : groups) M = std::max<uint32_t>(M, (uint32_t)g.second.size());
    const GemmArith ar = gemm_arith(ctx, gemm_weights_small(ctx, W, (size_t)O * K));
    const bool small = ar.small, mfma = gemm_mfma_ok(ctx, ar, M, K);
    // gather rows padded with -1 to 16 B multiples (+ 8 spare): the kernels read 4 at a time; matrix-core form: 32 entries per K step
    const uint32_t Kp = mfma ? ((K + 31) / 32) * 32 : ((K + 15) & ~15u) + 16;       // gather rows: 16 spare entries (the VALU kernels request up to 2 x 8 terms ahead)
    std::vector<int32_t> hidx((size_t)G * Kp, -1), hoidx((size_t)G * M, -1), hbidx((size_t)G * M, 0);
    std::vector<uint32_t> member((size_t)G * M, NONE);           // output index of (group, m)
    {
        uint32_t g = 0;
        for (auto &kv : groups) {
            memcpy(&hidx[(size_t)g * Kp], kv.first.data(), K * 4);
            for (uint32_t m = 0; m < kv.second.size(); m++) member[(size_t)g * M + m] = kv.second[m];
            g++;
        }
    }
    for (size_t x = 0; x < member.size(); x++) if (member[x] != NONE) { hoidx[x] = (int32_t)member[x]; if (BP) hbidx[x] = bias_idx[member[x]]; }   // relative to the output base
    P.O = O; P.K = K; P.Kp = Kp; P.G = G; P.M = M; P.small = small; P.has_bias = BP != nullptr; P.bias_pt = bias_pt; P.bias_count = BP ? BP->count : 0;
    P.two = ar.two; P.lazy = ar.lazy; P.mfma = mfma;
    std::vector<char> wbytes;
    auto row = [&](uint32_t g, uint32_t m) -> const uint64_t * { return member[(size_t)g * M + m] == NONE ? nullptr : W + (size_t)member[(size_t)g * M + m] * K; };
    if (mfma) {
        P.P = gemm_weight_planes(ctx, W, (size_t)O * K); P.mtiles = (M + 31) / 32; P.ksteps = (K + 31) / 32;
        pack_gemm_mfma(ctx, G, M, K, P.P, row, [&](uint32_t g, uint32_t kk) { return hidx[(size_t)g * Kp + kk] >= 0; }, wbytes);
    } else {
        auto tap = [&](uint32_t g, uint32_t kk) { return hidx[(size_t)g * Kp + kk] >= 0; };
        pack_gemm_weights(ctx, G, M, K, small, row, tap, P.MT, wbytes);
        P.one = gemm_one_limb(ctx, ar, G, M, K, row, tap);
    }
    P.off_oidx = al(hidx.size() * 4); P.off_bidx = P.off_oidx + al(hoidx.size() * 4); P.off_w = P.off_bidx + al(hbidx.size() * 4);
    P.host.assign(P.off_w + al(wbytes.size()), 0);
    memcpy(P.host.data(), hidx.data(), hidx.size() * 4);
    memcpy(P.host.data() + P.off_oidx, hoidx.data(), hoidx.size() * 4);
    memcpy(P.host.data() + P.off_bidx, hbidx.data(), hbidx.size() * 4);
    memcpy(P.host.data() + P.off_w, wbytes.data(), wbytes.size());
    return 0;
}
// tables: device image of P.host (scratch or the plan's own allocation)
static int run_gemm_plan(cn_ctx *ctx, const GemmPlan &P, const char *tables, Buffer *I, Buffer *OB, uint32_t oi) {
    if (!range_ok(OB, oi, P.O)) return fail(CN_ERR_ARG, "output index out of range");
    if (I == OB) return fail(CN_ERR_ARG, "scalar GEMM cannot run in place");
    if (P.max_in > I->count) return fail(CN_ERR_ARG, "input index out of range");
    // Evaluator::multiply_plain / add take ciphertexts of any size: size 3 = products that have not been relinearized yet (the sum of weighted
    // products is then relinearized once per OUTPUT instead of once per input)
    if (I->size != OB->size || I->size < 2 || I->size > 3) return fail(CN_ERR_ARG, "scalar GEMM: input and output ciphertext sizes must match (2 or 3)");
    const uint64_t *bias = nullptr;
    if (P.has_bias) {
        Buffer *BP = getbuf(ctx, P.bias_pt, 1);
        if (!BP || BP->count < P.bias_count) return fail(CN_ERR_ARG, "invalid bias plaintext handle");
        bias = BP->d;
    }
    GemmLaunch gl{P.small, P.two, false, P.MT, I->d, tables, tables + P.off_w, tables + P.off_oidx, bias, tables + P.off_bidx, OB->d,
                  P.G, P.M, P.K, P.lazy, P.Kp, oi, P.P, P.mtiles, P.ksteps, I->size, (uint32_t)ctx->gemm_order, P.one};
    CHECK(P.mfma ? cn_l_gemm_mfma(ctx, gl) : cn_l_gemm(ctx, gl));
    ctx->st.PlainMultiplication += P.nnz; ctx->st.Addition += P.nnz - P.O;
    if (P.has_bias) ctx->st.PlainAddition += P.O;
    return 0;
}
extern "C" int cn_scalar_gemm(cn_ctx *ctx, cn_handle in, const int32_t *idx, const uint64_t *W, uint32_t O, uint32_t K, cn_handle bias_pt,
                              const int32_t *bias_idx, cn_handle out, uint32_t oi) { API_BODY
    LOCK; GETCT(I, in, 0); GETCT(OB, out, 0);
    Buffer *BP = bias_pt ? getbuf(ctx, bias_pt, 1) : nullptr;
    GemmPlan P;
    CHECK(build_gemm_plan(ctx, idx, W, O, K, BP, bias_pt, bias_idx, P));
    CHECK(ensure_scratch(ctx, al(P.host.size())));
    char *tables; CHECK(upload_tmp(ctx, P.host.data(), P.host.size(), &tables));
    return run_gemm_plan(ctx, P, tables, I, OB, oi);
API_END }
// Plan once, apply per inference: the weight tiles and gather tables stay in HBM (cn_free releases the plan).
extern "C" int cn_gemm_plan_create(cn_ctx *ctx, const int32_t *idx, const uint64_t *W, uint32_t O, uint32_t K, cn_handle bias_pt, const int32_t *bias_idx,
                                   cn_handle *plan) { API_BODY
    LOCK; NOT_CAPTURING("cn_gemm_plan_create");
    if (!plan) return fail(CN_ERR_ARG, "null argument");
    Buffer *BP = bias_pt ? getbuf(ctx, bias_pt, 1) : nullptr;
    std::shared_ptr<GemmPlan> P = std::make_shared<GemmPlan>();
    CHECK(build_gemm_plan(ctx, idx, W, O, K, BP, bias_pt, bias_idx, *P));
    HIPCHK(hipMalloc((void **)&P->dev, P->host.size()));
    HIPCHK(hipMemcpy(P->dev, P->host.data(), P->host.size(), hipMemcpyHostToDevice));
    P->host.clear(); P->host.shrink_to_fit();
    Buffer b; b.kind = 2; b.count = O; b.size = 0; b.d = nullptr; b.item_words = 0; b.plan = P;
    *plan = ctx->bufs.insert(std::move(b));
    return 0;
API_END }
extern "C" int cn_gemm_plan_apply(cn_ctx *ctx, cn_handle plan, cn_handle in, cn_handle out, uint32_t oi) { API_BODY
    LOCK; GETCT(I, in, 0); GETCT(OB, out, 0);
    Buffer *PB = getbuf(ctx, plan, 2);
    if (!PB || !PB->plan) return fail(CN_ERR_ARG, "invalid scalar GEMM plan handle");
    return run_gemm_plan(ctx, *PB->plan, PB->plan->dev, I, OB, oi);
API_END }

// ---------------------------------------------------------------- BEHZ multiply / key switching
// tensor product fused into the inverse transform (register-radix sizes only); returns false when the caller must fall back
static bool run_intt_tensor(cn_ctx *c, const uint64_t *A, const uint64_t *B, uint64_t *D, uint32_t cnt, uint32_t base_off, uint32_t Lm) {
    if (c->legacy_ntt || c->hc.logn < 10 || c->hc.logn > 14) return false;
    bool f64 = c->use_f64, light = true;
    for (uint32_t m = base_off; m < base_off + Lm; m++) {
        f64 = f64 && c->hc.f64ok[m];
        uint64_t q = m < c->hc.k ? c->hc.q[m].q : c->hc.bsk[m - c->hc.k].q;
        if (q >> 44) light = false;
    }
    bool ok = rr_ops[f64 && light ? POL_F64L : (f64 ? POL_F64 : POL_U64)]->intt_tensor(c, A, B, D, cnt, base_off, Lm);
    if (ok) { launch_count(c); c->st.ntt_inverse_limbs += (uint64_t)cnt * 3 * Lm; }
    return ok;
}
// squaring: forward transforms, tensor and inverse transforms of one (ciphertext, limb) in ONE kernel (FP64 policies)
static bool square_fused_ok(cn_ctx *c, uint32_t base_off, uint32_t Lm, bool &light) {
    if (!c->sq_fused || c->legacy_ntt || !c->use_f64 || c->hc.logn < 10 || c->hc.logn > 14) return false;
    light = true;
    for (uint32_t m = base_off; m < base_off + Lm; m++) {
        if (!c->hc.f64ok[m]) return false;
        uint64_t q = m < c->hc.k ? c->hc.q[m].q : c->hc.bsk[m - c->hc.k].q;
        if (q >> 44) light = false;
    }
    return true;
}
static void run_square_fused(cn_ctx *c, const uint64_t *A, size_t astride, const uint64_t *const *atab, uint64_t *D, uint32_t cnt, uint32_t base_off, uint32_t Lm, bool light) {
    rr_ops[light ? POL_F64L : POL_F64]->square_fused(c, A, astride, atab, D, cnt, base_off, Lm);
    launch_count(c);
    c->st.ntt_forward_limbs += (uint64_t)cnt * 2 * Lm; c->st.ntt_inverse_limbs += (uint64_t)cnt * 3 * Lm;
}
// the context's second stream (squaring overlap): created on first use, kept only if it runs beside the context's own stream (a hardware queue of its own)
static bool aux_stream_ready(cn_ctx *ctx) {
    if (ctx->stream2) return true;
    if (ctx->stream2_failed) return false;
    hipStream_t cand[4] = {nullptr, nullptr, nullptr, nullptr};
    int got = -1;
    for (int i = 0; i < 4 && got < 0; i++) {
        if (hipStreamCreateWithFlags(&cand[i], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); cand[i] = nullptr; break; }
        if (!streams_share_a_queue(cand[i], ctx->stream)) got = i;
    }
    for (int i = 0; i < 4; i++) if (cand[i] && i != got) (void)hipStreamDestroy(cand[i]);
    if (got < 0 || hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (got >= 0) (void)hipStreamDestroy(cand[got]);
        ctx->stream2_failed = true;
        return false;
    }
    ctx->stream2 = cand[got];
    return true;
}
static size_t mul_scratch_per_ct(cn_ctx *c, bool square) {
    size_t n = c->hc.n, k = c->hc.k, kb = c->hc.kb;
    size_t w = (square ? 1 : 2) * 2 * (k + kb) * n + 3 * (k + kb) * n;
    return al(w * 8) + 1024;
}
// a, b: pointers to first operand ciphertext (size 2); out3: [cnt][3][k][N]; scratch must be ensured by caller.  atab / btab: one
// operand address per ciphertext instead of a + ct*astride*ctw (deferred per-ciphertext calls; atab == btab: squarings)
static int do_multiply(cn_ctx *ctx, const uint64_t *a, uint32_t astride, const uint64_t *b, uint32_t bstride, uint64_t *out3, uint32_t cnt,
                       const uint64_t *const *atab = nullptr, const uint64_t *const *btab = nullptr) {
    const uint32_t n = ctx->hc.n, k = ctx->hc.k, kb = ctx->hc.kb;
    const bool square = atab ? atab == btab : (a == b && astride == bstride);
    // squarings on the FP64 path: one fused kernel per base does forward transforms, tensor and inverse transforms; its q side reads the
    // input ciphertexts in place, so k_behz_extend only has to produce the Bsk limbs
    bool lq = false, lb = false;
    const bool fused = square && ctx->hc.behz_f64 && square_fused_ok(ctx, 0, k, lq) && square_fused_ok(ctx, k, kb, lb);
    uint64_t *aq = fused ? nullptr : salloc<uint64_t>(ctx, (size_t)cnt * 2 * k * n), *ab = salloc<uint64_t>(ctx, (size_t)cnt * 2 * kb * n);
    uint64_t *bq = aq, *bb = ab;
    if (!square) { bq = salloc<uint64_t>(ctx, (size_t)cnt * 2 * k * n); bb = salloc<uint64_t>(ctx, (size_t)cnt * 2 * kb * n); }
    uint64_t *dq = salloc<uint64_t>(ctx, (size_t)cnt * 3 * k * n), *db = salloc<uint64_t>(ctx, (size_t)cnt * 3 * kb * n);
    if ((!fused && !aq) || !ab || (!square && (!bq || !bb)) || !dq || !db) return fail(CN_ERR_HIP, "internal: scratch exhausted in multiply");
    // Squaring of a batch, "sq_overlap": the q-side transform kernel needs only the input, the Bsk side needs k_behz_extend's output - so the q side runs on a second
    // stream of the context beside [extend -> Bsk side] and joins in front of k_behz_floor (round 6).  The two resident transform kernels cannot share a CU
    // (130 KiB of LDS each), but the HBM-bound base extension (no LDS, few registers) runs beside the q side's workgroups instead of in front of them.
    const bool overlap = fused && ctx->sq_overlap && !ctx->capturing && cnt >= 64 && aux_stream_ready(ctx);
    if (overlap) {
        HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        std::swap(ctx->stream, ctx->stream2);
        run_square_fused(ctx, a, (size_t)astride * 2 * k * n, atab, dq, cnt, 0, k, lq);
        std::swap(ctx->stream, ctx->stream2);
        HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
    }
    CHECK(cn_l_behz_extend(ctx, a, astride, atab, aq, ab, cnt));
    if (!square) CHECK(cn_l_behz_extend(ctx, b, bstride, btab, bq, bb, cnt));
    if (fused) {
        if (!overlap) run_square_fused(ctx, a, (size_t)astride * 2 * k * n, atab, dq, cnt, 0, k, lq);
        run_square_fused(ctx, ab, (size_t)2 * kb * n, nullptr, db, cnt, k, kb, lb);
        if (overlap) HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    } else {
    CHECK(cn_run_ntt(ctx, aq, cnt * 2 * k, 0, k, 0)); CHECK(cn_run_ntt(ctx, ab, cnt * 2 * kb, k, kb, 0));
    if (!square) { CHECK(cn_run_ntt(ctx, bq, cnt * 2 * k, 0, k, 0)); CHECK(cn_run_ntt(ctx, bb, cnt * 2 * kb, k, kb, 0)); }
    if (!run_intt_tensor(ctx, aq, bq, dq, cnt, 0, k) || !run_intt_tensor(ctx, ab, bb, db, cnt, k, kb)) {
        hipLaunchKernelGGL(k_tensor, dim3(cnt * k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, aq, bq, dq, ctx->dc, ctx->chunks, k, 0u);
        hipLaunchKernelGGL(k_tensor, dim3(cnt * kb * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, ab, bb, db, ctx->dc, ctx->chunks, kb, k);
        HIPCHK(hipGetLastError()); launch_count(ctx, 2);
        CHECK(cn_run_ntt(ctx, dq, cnt * 3 * k, 0, k, 1)); CHECK(cn_run_ntt(ctx, db, cnt * 3 * kb, k, kb, 1));
    }
    }
    HIPCHK(hipGetLastError());
    CHECK(cn_l_behz_floor(ctx, dq, db, out3, cnt));
    ctx->st.Multiplication += cnt;
    return 0;
}
template <int EPT> static void launch_ks_legacy(cn_ctx *c, uint32_t nt, const KsArgs &a) {
    hipLaunchKernelGGL(k_keyswitch<EPT>, dim3(a.cnt * c->hc.k), dim3(nt), (size_t)c->hc.n * 8, c->stream, a.target, a.tstride, a.add0, a.add1, a.astride, a.key,
                       a.out, c->dc, a.galois, a.out_tab);
}
static int ensure_ks_part(cn_ctx *ctx, size_t need) {
    if (need <= ctx->ks_part_cap) return 0;
    if (ctx->capturing || ctx->graphs_alive) return fail(CN_ERR_ARG, "the key-switch arena would have to grow while a graph is recorded / alive: run the sequence once before cn_graph_begin");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->ks_part) HIPCHK(hipFree(ctx->ks_part));
    ctx->ks_part = nullptr; ctx->ks_part_cap = 0;
    HIPCHK(hipMalloc(&ctx->ks_part, need));
    ctx->ks_part_cap = need;
    return 0;
}
// auto: the fused kernel runs cnt*k workgroups.  Up to 32 of them (1-6 ciphertexts) every digit gets its own workgroup; up to 160
// every source limb does; above that the fused kernel fills the chip by itself.
static uint32_t ks_digit_max_blocks() {              // (ciphertext, limb) blocks up to which the two-launch key switch runs one workgroup per DIGIT (above: per source limb); CN_KS_DIGIT_MAX overrides (A/B)
    // 10 since round 3 (1-2 ciphertexts; 32 before): with the four chains of an image on four hardware queues, per-source-limb workgroups cost the
    // chip less for 3-6 ciphertexts too (four chains 5.13-5.24 vs 5.36 ms per image, one chain alone unchanged; 50 / 65: 6.4 / 7.5 ms)
    static const uint32_t v = getenv("CN_KS_DIGIT_MAX") ? (uint32_t)atoi(getenv("CN_KS_DIGIT_MAX")) : 10u;
    return v;
}
#define KS_DIGIT_MAX_BLOCKS ks_digit_max_blocks()
static uint32_t ks_wide_max_blocks() {               // (ciphertext, limb) blocks up to which a key switch runs as two launches; CN_KS_WIDE_MAX overrides (A/B)
    static const uint32_t v = getenv("CN_KS_WIDE_MAX") ? (uint32_t)atoi(getenv("CN_KS_WIDE_MAX")) : 160u;
    return v;
}
#define KS_WIDE_MAX_BLOCKS ks_wide_max_blocks()
// the variant do_keyswitch takes for `cnt` ciphertexts: 0 = the fused kernel, 1 / 2 = two launches (KsArgs::mode)
static int ks_planned_mode(cn_ctx *ctx, uint32_t cnt, int galois) {
    const uint32_t k = ctx->hc.k, tot_dig = galois ? ctx->hc.gk_tot : ctx->hc.rl_tot;
    const bool rr = !ctx->legacy_ntt && ctx->hc.logn >= 10 && ctx->hc.logn <= 14;
    if (!(rr && (ctx->ks_wide > 0 || (ctx->ks_wide < 0 && cnt * k <= KS_WIDE_MAX_BLOCKS)))) return 0;
    // N = 16384: 1024-thread workgroups cannot hold two accumulator sets without spilling -> per-digit only
    const int mode = ctx->ks_wide == 2 || (ctx->ks_wide < 0 && cnt * k > KS_DIGIT_MAX_BLOCKS && ctx->hc.logn < 14) ? 2 : 1;
    return (size_t)cnt * (mode == 2 ? k : tot_dig) * ctx->ctw2 * 8 > ctx->smax ? 0 : mode;
}
// perm_elt != 0 (two-launch variants only - the caller asks ks_planned_mode first): target / add0 are the c1 / c0 of the ciphertext a rotation
// READS and the kernels apply the automorphism x -> x^perm_elt while loading them
// N = 16384, fused path: one launch per key switch (k_keyswitch_pair14).  A rotation then hands in target = sigma(c1) (permuted ahead of time: k_galois_limbs, or the
// previous link of a rotate-and-add chain), add0 = the unpermuted c0 and perm_elt; next_elt / next_out ask for sigma_next of the new c1 on the side.
static bool ks_pair14_ok(cn_ctx *ctx, uint32_t cnt, int galois, const KsKey &key);
static int do_keyswitch(cn_ctx *ctx, const uint64_t *target, size_t tstride, const uint64_t *add0, const uint64_t *add1, size_t astride,
                        const KsKey &key, uint64_t *out, uint32_t cnt, int galois, const uint64_t *extra = nullptr, size_t xstride = 0,
                        uint64_t *const *out_tab = nullptr, uint32_t perm_elt = 0, const KsItem *items = nullptr, uint32_t next_elt = 0, uint64_t *next_out = nullptr) {
    const uint32_t n = ctx->hc.n, k = ctx->hc.k, tot_dig = galois ? ctx->hc.gk_tot : ctx->hc.rl_tot;
    uint64_t qmax = 0; for (uint32_t j = 0; j < k; j++) qmax = std::max(qmax, ctx->hc.q[j].q);
    const int bits = 64 - __builtin_clzll(qmax);
    KsArgs a{target, tstride, add0, add1, astride, key.d, out, cnt, galois, extra, xstride,
             key.f64 ? (bits >= 50 ? 1u : (1u << std::min(10, 50 - bits))) : 0xffffffffu,     // lazy FP64 accumulators: |term| <= 2.1 q, sum below 2^52
             0, out_tab};
    if (ctx->ks_xcd == 1) a.xcd_cts = cnt & ~7u;
    else if (ctx->ks_xcd == 2) a.xcd_cts = 0x80000000u;
    const bool rr = !ctx->legacy_ntt && ctx->hc.logn >= 10 && ctx->hc.logn <= 14;                // register-radix kernels available
    a.mode = ks_planned_mode(ctx, cnt, galois);
    if (a.mode) CHECK(ensure_ks_part(ctx, (size_t)cnt * (a.mode == 2 ? k : tot_dig) * ctx->ctw2 * 8));
    a.perm_elt = perm_elt; a.items = items;
    a.next_elt = next_elt; a.next_out = next_out;
    const bool pair = ks_pair14_ok(ctx, cnt, galois, key);
    if ((perm_elt || items || next_elt) && !a.mode && !(pair && !items)) return fail(CN_ERR_ARG, "internal: automorphism inside the fused key switch");
    if (pair) {                                                                                      // N = 16384: both 8192-point halves of a limb in one workgroup, one launch
        CHECK(ensure_ks_part(ctx, (size_t)cnt * ctx->ctw2 * 8));
        a.xcd_cts = ctx->ks_xcd == 1 ? (cnt & ~7u) : 0u;
        ks_ops[bits <= 44 ? POL_F64L : POL_F64]->pair14(ctx, a);
    } else if (a.mode == 0 && rr && key.f64 && ctx->hc.logn == 14 && ctx->hc.twdh && ctx->ks_split14) {   // ... as two workgroups per limb + a combining pass (rounds 1-4; A/B)
        CHECK(ensure_ks_part(ctx, (size_t)cnt * ctx->ctw2 * 8));
        ks_ops[bits <= 44 ? POL_F64L : POL_F64]->split14(ctx, a);
        hipLaunchKernelGGL(k_ks_combine14, dim3(cnt * 2 * k * (n / 512)), dim3(256), 0, ctx->stream, (const uint64_t *)ctx->ks_part, add0, add1, astride, out, ctx->dc,
                           extra, xstride, out_tab);
        launch_count(ctx);
    } else {
        bool done = false;
        if (key.f64) {
            done = rr && ks_ops[bits <= 44 ? POL_F64L : POL_F64]->launch(ctx, a);
            if (!done) return fail(CN_ERR_ARG, "internal: FP64 key without FP64 kernel");
        } else if (rr) done = ks_ops[POL_U64]->launch(ctx, a);
        if (!done) {                          // radix-2 LDS fallback (N < 1024, legacy_ntt): no fused accumulator -> one element-wise add behind it
            const uint32_t nt = std::min<uint32_t>(1024, n);
            switch (n / nt) {
                case 1: launch_ks_legacy<1>(ctx, nt, a); break;
                case 2: launch_ks_legacy<2>(ctx, nt, a); break;
                case 4: launch_ks_legacy<4>(ctx, nt, a); break;
                case 8: launch_ks_legacy<8>(ctx, nt, a); break;
                case 16: launch_ks_legacy<16>(ctx, nt, a); break;
                default: return fail(CN_ERR_ARG, "unsupported poly modulus degree for key switching");
            }
            if (extra) {
                if (xstride != ctx->ctw2 || out_tab) return fail(CN_ERR_ARG, "internal: accumulator stride");
                hipLaunchKernelGGL(k_addsub, dim3(cnt * 2 * k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, out, extra, out, ctx->dc, ctx->chunks, 0);
                launch_count(ctx);
            }
        }
    }
    HIPCHK(hipGetLastError()); launch_count(ctx);
    ctx->st.ntt_forward_limbs += (uint64_t)cnt * tot_dig * k; ctx->st.ntt_inverse_limbs += (uint64_t)cnt * 2 * k;
    return 0;
}
static bool ks_pair14_ok(cn_ctx *ctx, uint32_t cnt, int galois, const KsKey &key) {
    return ctx->ks_pair14 && ctx->ks_split14 && !ctx->legacy_ntt && ctx->hc.logn == 14 && ctx->hc.twdh && key.f64 && ks_planned_mode(ctx, cnt, galois) == 0;
}
static uint32_t chunk_for(cn_ctx *ctx, size_t per_ct, uint32_t count) {
    size_t c = std::max<size_t>(1, ctx->smax / per_ct);
    return (uint32_t)std::min<size_t>(c, count);
}

extern "C" int cn_multiply(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out3, uint32_t oi, uint32_t count) { API_BODY
    LOCK; GETCT(A, a, 2); GETCT(B, b, 2); GETCT(O, out3, 3);
    if (!range_ok(A, ai, count) || !range_ok(B, bi, count) || !range_ok(O, oi, count)) return fail(CN_ERR_ARG, "index out of range");
    if (!count) return 0;
    const uint64_t *pa = A->d + ai * A->item_words, *pb = B->d + bi * B->item_words;
    size_t per = mul_scratch_per_ct(ctx, pa == pb);
    uint32_t ch = chunk_for(ctx, per, count);
    for (uint32_t s = 0; s < count; s += ch) {
        uint32_t c = std::min(ch, count - s);
        CHECK(ensure_scratch(ctx, per * c + 4096));
        CHECK(do_multiply(ctx, pa + s * A->item_words, 1, pb + s * B->item_words, 1, O->d + (oi + s) * O->item_words, c));
    }
    return 0;
API_END }
extern "C" int cn_relinearize(cn_ctx *ctx, cn_handle in3, uint32_t ii, cn_handle out, uint32_t oi, uint32_t count) { API_BODY
    LOCK; GETCT(I, in3, 3); GETCT(O, out, 2);
    if (!range_ok(I, ii, count) || !range_ok(O, oi, count)) return fail(CN_ERR_ARG, "index out of range");
    if (!ctx->rlk.d) return fail(CN_ERR_NOKEY, "relinearization keys not set");
    if (!count) return 0;
    const size_t kn = (size_t)ctx->hc.k * ctx->hc.n;
    const uint64_t *p = I->d + ii * I->item_words;
    CHECK(do_keyswitch(ctx, p + 2 * kn, 3 * kn, p, p + kn, 3 * kn, ctx->rlk, O->d + oi * O->item_words, count, 0));
    ctx->st.Relinarization += count;
    return 0;
API_END }
static int mul_relin_body(cn_ctx *ctx, cn_handle a, uint32_t ai, uint32_t astride, cn_handle b, uint32_t bi, uint32_t bstride, cn_handle out, uint32_t oi, uint32_t count);
extern "C" int cn_mul_relin(cn_ctx *ctx, cn_handle a, uint32_t ai, uint32_t astride, cn_handle b, uint32_t bi, uint32_t bstride, cn_handle out,
                            uint32_t oi, uint32_t count) {
    if (submit_async(ctx) && count <= DEFER_STAGED_MAX) return ring_push(ctx, SUB_MUL_RELIN, count, a, ai, b, bi, out, oi, astride, bstride);      // PointwiseMultiply of one column
    API_BODY LOCK_ONLY; return mul_relin_body(ctx, a, ai, astride, b, bi, bstride, out, oi, count); API_END
}
static int mul_relin_body(cn_ctx *ctx, cn_handle a, uint32_t ai, uint32_t astride, cn_handle b, uint32_t bi, uint32_t bstride, cn_handle out, uint32_t oi, uint32_t count) {
    if (deferring(ctx)) return defer_mul_relin(ctx, a, ai, astride, b, bi, bstride, out, oi, count);
    CHECK(cn_defer_flush(ctx));
    GETCT(A, a, 2); GETCT(B, b, 2); GETCT(O, out, 2);
    if (!range_ok(A, ai, astride ? count : 1, astride ? astride : 1) || !range_ok(B, bi, bstride ? count : 1, bstride ? bstride : 1) || !range_ok(O, oi, count))
        return fail(CN_ERR_ARG, "index out of range");
    if (!ctx->rlk.d) return fail(CN_ERR_NOKEY, "relinearization keys not set");
    if (!count) return 0;
    const size_t kn = (size_t)ctx->hc.k * ctx->hc.n;
    const uint64_t *pa = A->d + ai * A->item_words, *pb = B->d + bi * B->item_words;
    const bool square = (pa == pb && astride == bstride);
    size_t per = mul_scratch_per_ct(ctx, square) + al(3 * kn * 8);
    uint32_t ch = chunk_for(ctx, per, count);
    for (uint32_t s = 0; s < count; s += ch) {
        uint32_t c = std::min(ch, count - s);
        CHECK(ensure_scratch(ctx, per * c + 8192));
        uint64_t *t3 = salloc<uint64_t>(ctx, (size_t)c * 3 * kn);
        CHECK(do_multiply(ctx, pa + (size_t)s * astride * A->item_words, astride, pb + (size_t)s * bstride * B->item_words, bstride, t3, c));
        CHECK(do_keyswitch(ctx, t3 + 2 * kn, 3 * kn, t3, t3 + kn, 3 * kn, ctx->rlk, O->d + (oi + s) * O->item_words, c, 0));
    }
    ctx->st.Relinarization += count;
    return 0;
}

// ---------------------------------------------------------------- rotations
// in/out device pointers to size-2 ciphertext arrays; tmp holds count size-2 ciphertexts
// acc != nullptr: out = acc + galois(in) in the same launches (acc may alias out and/or in)
// pre: sigma_elt(c1) of `in` if somebody has produced it already ([ct][k][N]); next_elt / next_out: see do_keyswitch (both only on the one-launch N = 16384 path)
static int do_galois(cn_ctx *ctx, const uint64_t *in, uint64_t elt, uint64_t *out, uint64_t *tmp, uint32_t count, const uint64_t *acc = nullptr,
                     const uint64_t *pre = nullptr, uint64_t next_elt = 0, uint64_t *next_out = nullptr) {
    auto it = ctx->gk.find(elt);
    if (it == ctx->gk.end() || !it->second.d) return fail(CN_ERR_NOKEY, "Galois key not present");

    const size_t kn = (size_t)ctx->hc.k * ctx->hc.n;
    uint32_t limbs = count * 2 * ctx->hc.k;
    // (the one-launch kernel reads c0 of `in` while other workgroups already write `out`: safe when the two arrays are the same or disjoint - a workgroup only touches
    // its own (ciphertext, limb) - not when they overlap with a shift: those calls keep the permutation pass, which has read all of `in` before anything is written)
    const bool shifted = in != out && in < out + (size_t)count * ctx->ctw2 && out < in + (size_t)count * ctx->ctw2;
    const bool acc_shifted = acc && acc != out && acc < out + (size_t)count * ctx->ctw2 && out < acc + (size_t)count * ctx->ctw2;
    if (acc_shifted) return fail(CN_ERR_ARG, "rotate-and-add: accumulator and result ranges overlap partially (use the same range or disjoint ranges)");
    if (!shifted && ks_pair14_ok(ctx, count, 1, it->second)) {           // N = 16384, batch: c1 is permuted once (here unless the caller brings it), c0 inside the key switch
        if (!pre) {
            hipLaunchKernelGGL(k_galois_limbs, dim3(count * ctx->hc.k), dim3(1024), (size_t)ctx->hc.n * 8, ctx->stream, in + kn, 2 * kn, tmp, kn, ctx->dc, elt);
            HIPCHK(hipGetLastError()); launch_count(ctx);
            pre = tmp;
        }
        CHECK(do_keyswitch(ctx, pre, kn, in, nullptr, 2 * kn, it->second, out, count, 1, acc, ctx->ctw2, nullptr, (uint32_t)elt, nullptr, (uint32_t)next_elt, next_out));
        ctx->st.Rotation += count;
        if (acc) ctx->st.Addition += count;
        return 0;
    }
    if (pre || next_elt) return fail(CN_ERR_ARG, "internal: rotation chain outside the one-launch key switch");
    // small batches (two-launch key switch): no permutation pass - the key-switch kernels apply the automorphism while they load c1 and c0
    if (!shifted && ctx->ks_perm_fused && ks_planned_mode(ctx, count, 1) != 0) {
        CHECK(do_keyswitch(ctx, in + kn, 2 * kn, in, nullptr, 2 * kn, it->second, out, count, 1, acc, ctx->ctw2, nullptr, (uint32_t)elt));
        ctx->st.Rotation += count;
        if (acc) ctx->st.Addition += count;
        return 0;
    }
    if (ctx->hc.n >= 1024) hipLaunchKernelGGL(k_galois_lds, dim3(limbs), dim3(std::min<uint32_t>(1024, ctx->hc.n / 4)), (size_t)ctx->hc.n * 8, ctx->stream, in, tmp, ctx->dc, elt);
    else hipLaunchKernelGGL(k_galois, dim3(limbs * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, in, tmp, ctx->dc, ctx->chunks, elt);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    CHECK(do_keyswitch(ctx, tmp + kn, 2 * kn, tmp, nullptr, 2 * kn, it->second, out, count, 1, acc, ctx->ctw2));
    ctx->st.Rotation += count;
    if (acc) ctx->st.Addition += count;
    return 0;
}
// A rotation whose result range overlaps its operand range with a SHIFT (the same handle, different first indices): the kernels that apply the automorphism while they
// load (one-launch N = 16384 kernel, two-launch small-batch kernels) would read ciphertext c after another workgroup has written c' over it - such calls take the
// permutation pass (do_galois: `shifted`), which has read the whole operand before anything is written (they used to be refused on every path).  Only the
// rotate-and-ADD forms still refuse a partially overlapping ACCUMULATOR: no path reads it ahead of the stores.
static bool shifted_overlap(const Buffer *I, uint32_t ii, const Buffer *O, uint32_t oi, uint32_t count) { return I == O && ii != oi && ii < oi + count && oi < ii + count; }
static int galois_impl(cn_ctx *ctx, Buffer *I, uint32_t ii, uint64_t elt, Buffer *O, uint32_t oi, uint32_t count) {
    if (!range_ok(I, ii, count) || !range_ok(O, oi, count)) return fail(CN_ERR_ARG, "index out of range");
    if (!count) return 0;
    CHECK(ensure_scratch(ctx, al(count * ctx->ctw2 * 8)));
    uint64_t *tmp = salloc<uint64_t>(ctx, count * ctx->ctw2);
    return do_galois(ctx, I->d + ii * I->item_words, elt, O->d + oi * O->item_words, tmp, count);
}
static bool galois_key_present(cn_ctx *ctx, uint64_t elt) { auto it = ctx->gk.find(elt); return it != ctx->gk.end() && it->second.d; }
extern "C" int cn_apply_galois(cn_ctx *ctx, cn_handle in, uint32_t ii, uint64_t elt, cn_handle out, uint32_t oi, uint32_t count) { API_BODY
    LOCK; GETCT(I, in, 2); GETCT(O, out, 2);
    return galois_impl(ctx, I, ii, elt, O, oi, count);
API_END }
// Evaluator::rotate_internal: direct key if present, otherwise non-adjacent-form decomposition
static bool has_direct_key(cn_ctx *ctx, int steps) {
    uint64_t elt = cn_galois_elt_from_step(ctx, steps);
    auto it = ctx->gk.find(elt);
    return elt && it != ctx->gk.end() && it->second.d;
}
static int rotate_rec(cn_ctx *ctx, uint64_t *cur, int steps, uint64_t *tmp, uint32_t count) {
    if (steps == 0) return 0;
    uint64_t elt = cn_galois_elt_from_step(ctx, steps);
    if (!elt) return fail(CN_ERR_ARG, "step count too large");
    auto it = ctx->gk.find(elt);
    if (it != ctx->gk.end() && it->second.d) return do_galois(ctx, cur, elt, cur, tmp, count);
    std::vector<int> naf;
    bool sign = steps < 0; int v = std::abs(steps);
    for (int i = 0; v; i++) { int zi = (v & 1) ? 2 - (v & 3) : 0; v = (v - zi) >> 1; if (zi) naf.push_back((sign ? -zi : zi) * (1 << i)); }
    if (naf.size() == 1) return fail(CN_ERR_NOKEY, "Galois key not present");
    for (int s : naf) {
        if ((uint32_t)std::abs(s) == ctx->hc.n / 2) continue;
        CHECK(rotate_rec(ctx, cur, s, tmp, count));
    }
    return 0;
}
static int rotate_rows_impl(cn_ctx *ctx, Buffer *I, uint32_t ii, int steps, Buffer *O, uint32_t oi, uint32_t count) {
    if (!range_ok(I, ii, count) || !range_ok(O, oi, count)) return fail(CN_ERR_ARG, "index out of range");
    if (!count) return 0;
    const bool shifted = shifted_overlap(I, ii, O, oi, count);
    CHECK(ensure_scratch(ctx, al(count * ctx->ctw2 * 8) * (shifted ? 2 : 1)));
    uint64_t *tmp = salloc<uint64_t>(ctx, count * ctx->ctw2);
    uint64_t *o = O->d + oi * O->item_words; const uint64_t *i = I->d + ii * I->item_words;
    if (steps != 0 && has_direct_key(ctx, steps)) return do_galois(ctx, i, cn_galois_elt_from_step(ctx, steps), o, tmp, count);   // one hop: no staging copy
    if (shifted) {                                    // overlapping ranges: through a staging array (a device-to-device copy between overlapping ranges is undefined)
        uint64_t *stage = salloc<uint64_t>(ctx, count * ctx->ctw2);
        HIPCHK(hipMemcpyAsync(stage, i, count * ctx->ctw2 * 8, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(o, stage, count * ctx->ctw2 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    } else if (o != i) HIPCHK(hipMemcpyAsync(o, i, count * ctx->ctw2 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return rotate_rec(ctx, o, steps, tmp, count);
}
// can RotateRows(steps) run with the keys this context holds (direct key, or every hop of the NAF decomposition)?  Queued rotations are
// checked when they are queued, like every other argument.
static int rotate_check(cn_ctx *ctx, int steps) {
    if (steps == 0) return 0;
    const uint64_t elt = cn_galois_elt_from_step(ctx, steps);
    if (!elt) return fail(CN_ERR_ARG, "step count too large");
    if (galois_key_present(ctx, elt)) return 0;
    std::vector<int> naf;
    bool sign = steps < 0; int v = std::abs(steps);
    for (int i = 0; v; i++) { int zi = (v & 1) ? 2 - (v & 3) : 0; v = (v - zi) >> 1; if (zi) naf.push_back((sign ? -zi : zi) * (1 << i)); }
    if (naf.size() == 1) return fail(CN_ERR_NOKEY, "Galois key not present");
    for (int s2 : naf) { if ((uint32_t)std::abs(s2) == ctx->hc.n / 2) continue; CHECK(rotate_check(ctx, s2)); }
    return 0;
}
// ---- rotations of n ciphertexts by n DIFFERENT step counts as one launch chain (cn_rotate_rows_many; the queued RotateRows calls of one level).
// A single-image network rotates the 13 masked vectors of an Interleave by 13 different amounts, the 5 maps of a Vectorize by 5: one rotation
// is 2 dependent dispatches per hop, and dependent dispatches are what the latency of such a chain is made of.  The hops of a
// rotation (the element of its step count if the key exists, else its NAF terms - the same decomposition rotate_rec walks, so the words are the
// same) are taken in rounds: round r is ONE two-launch key switch over every ciphertext that has an r-th hop, each with its own key and element
// from a table (KsItem); round 0 reads the source and writes the destination, later rounds work on the destination in place.
struct Tab2 { const NTT_GLOBAL uint64_t *src; NTT_GLOBAL uint64_t *dst; };          // one (source, destination) pair of k_copy_tab
static int ensure_stage(cn_ctx *ctx, size_t bytes);
static int copy_by_table(cn_ctx *ctx, const std::vector<Tab2> &tab, Tab2 *dtab, uint32_t words_per_item);
struct RotJob { const uint64_t *src; uint64_t *dst; int steps; std::vector<uint64_t> elts; };
static int rotation_hops(cn_ctx *ctx, int steps, std::vector<uint64_t> &elts) {
    if (steps == 0) return 0;
    const uint64_t elt = cn_galois_elt_from_step(ctx, steps);
    if (!elt) return fail(CN_ERR_ARG, "step count too large");
    if (galois_key_present(ctx, elt)) { elts.push_back(elt); return 0; }
    std::vector<int> naf;
    bool sign = steps < 0; int v = std::abs(steps);
    for (int i = 0; v; i++) { int zi = (v & 1) ? 2 - (v & 3) : 0; v = (v - zi) >> 1; if (zi) naf.push_back((sign ? -zi : zi) * (1 << i)); }
    if (naf.size() == 1) return fail(CN_ERR_NOKEY, "Galois key not present");
    for (int s2 : naf) { if ((uint32_t)std::abs(s2) == ctx->hc.n / 2) continue; CHECK(rotation_hops(ctx, s2, elts)); }
    return 0;
}
static int rotate_jobs(cn_ctx *ctx, std::vector<RotJob> &jobs) {
    const uint32_t n = (uint32_t)jobs.size();
    if (!n) return 0;
    size_t rounds = 0;
    for (RotJob &j : jobs) { CHECK(rotation_hops(ctx, j.steps, j.elts)); rounds = std::max(rounds, j.elts.size()); }
    bool aliased = false;
    for (uint32_t a = 0; a < n && !aliased; a++) for (uint32_t b = 0; b < n; b++) if (a != b && (jobs[a].dst == jobs[b].src || jobs[a].dst == jobs[b].dst)) { aliased = true; break; }
    bool tables_ok = ctx->ks_perm_fused && !aliased && ks_planned_mode(ctx, n, 1) != 0;
    if (!tables_ok && !aliased && ctx->ks_perm_fused && n > 1) {
        // More rotations than ONE table-driven two-launch key switch takes (LoLa-CIFAR's ConvertToColumnVector: 83 maps at N = 16384 - 664 (ciphertext, limb)
        // blocks against the 160 up to which a key switch runs as two launches): pieces of the largest size that does, each a launch chain of its own, instead
        // of 83 x ~4 single-ciphertext rotations of two launches each (round 5: 632 -> ~40 launches per plaintext prime and image).  Independent jobs: any order.
        uint32_t piece = 0;
        for (uint32_t c = std::min<uint32_t>(n - 1, 64); c >= 2; c--) if (ks_planned_mode(ctx, c, 1) != 0) { piece = c; break; }
        if (piece) {
            for (uint32_t s0 = 0; s0 < n; s0 += piece) {
                std::vector<RotJob> part(jobs.begin() + s0, jobs.begin() + std::min<uint32_t>(n, s0 + piece));
                for (RotJob &j : part) j.elts.clear();
                CHECK(rotate_jobs(ctx, part));
            }
            return 0;
        }
    }
    if (!tables_ok) {                                      // large batches (fused kernel), aliased operands: one after the other
        CHECK(ensure_scratch(ctx, al(ctx->ctw2 * 8)));
        for (RotJob &j : jobs) {
            ctx->soff = 0;
            uint64_t *tmp = salloc<uint64_t>(ctx, ctx->ctw2);
            if (j.elts.size() == 1) { CHECK(do_galois(ctx, j.src, j.elts[0], j.dst, tmp, 1)); continue; }
            if (j.dst != j.src) HIPCHK(hipMemcpyAsync(j.dst, j.src, ctx->ctw2 * 8, hipMemcpyDeviceToDevice, ctx->stream));
            for (uint64_t e : j.elts) CHECK(do_galois(ctx, j.dst, e, j.dst, tmp, 1));
        }
        return 0;
    }
    {   // rotations by 0 steps: copies
        std::vector<Tab2> cp;
        for (RotJob &j : jobs) if (j.elts.empty() && j.dst != j.src) cp.push_back({(const NTT_GLOBAL uint64_t *)j.src, (NTT_GLOBAL uint64_t *)j.dst});
        if (!cp.empty()) { CHECK(ensure_stage(ctx, al(cp.size() * sizeof(Tab2)))); CHECK(copy_by_table(ctx, cp, (Tab2 *)ctx->stage, (uint32_t)ctx->ctw2)); }
    }
    for (size_t r = 0; r < rounds; r++) {
        std::vector<KsItem> items; std::vector<uint64_t *> outs;
        const KsKey *any = nullptr;
        for (RotJob &j : jobs) {
            if (j.elts.size() <= r) continue;
            const KsKey &key = ctx->gk.find(j.elts[r])->second;
            any = &key;
            items.push_back({r == 0 ? j.src : j.dst, key.d, (uint32_t)j.elts[r], 0u});
            outs.push_back(j.dst);
        }
        const uint32_t cnt = (uint32_t)items.size();
        ctx->soff = 0;
        CHECK(ensure_scratch(ctx, al(cnt * sizeof(KsItem)) + al(cnt * sizeof(uint64_t *))));
        KsItem *d_items = salloc<KsItem>(ctx, cnt);
        uint64_t **d_outs = salloc<uint64_t *>(ctx, cnt);
        const void *p_items, *p_outs;
        CHECK(place_table(ctx, items.data(), cnt * sizeof(KsItem), d_items, &p_items));
        CHECK(place_table(ctx, outs.data(), cnt * sizeof(uint64_t *), d_outs, &p_outs));
        CHECK(do_keyswitch(ctx, nullptr, 0, nullptr, nullptr, 0, *any, nullptr, cnt, 1, nullptr, 0, (uint64_t *const *)p_outs, 0, (const KsItem *)p_items));
        ctx->st.Rotation += cnt;
    }
    return 0;
}

extern "C" int cn_rotate_rows(cn_ctx *ctx, cn_handle in, uint32_t ii, int steps, cn_handle out, uint32_t oi, uint32_t count) { API_BODY
    LOCK_ONLY; GETCT(I, in, 2); GETCT(O, out, 2);
    if (deferring(ctx) && count && count <= DEFER_STAGED_MAX) {
        if (!range_ok(I, ii, count) || !range_ok(O, oi, count)) return fail(CN_ERR_ARG, "index out of range");
        CHECK(rotate_check(ctx, steps));
        return defer_staged(ctx, DOP_ROT, I, ii, nullptr, 0, nullptr, 0, O, oi, count, steps);
    }
    CHECK(cn_defer_flush(ctx));
    return rotate_rows_impl(ctx, I, ii, steps, O, oi, count);
API_END }
// RotateRows of n ciphertexts by n different step counts, one launch chain (see include/cnhip.h)
extern "C" int cn_rotate_rows_many(cn_ctx *ctx, cn_handle in, const uint32_t *ii, const int *steps, uint32_t n, cn_handle out, const uint32_t *oi) { API_BODY
    LOCK_ONLY; GETCT(I, in, 2); GETCT(O, out, 2);
    if (!n) return 0;
    if (!ii || !steps || !oi) return fail(CN_ERR_ARG, "null argument");
    for (uint32_t i = 0; i < n; i++) {
        if (!range_ok(I, ii[i], 1) || !range_ok(O, oi[i], 1)) return fail(CN_ERR_ARG, "index out of range");
        CHECK(rotate_check(ctx, steps[i]));
        for (uint32_t j = 0; j < i; j++) if (O == I ? (oi[i] == oi[j] || oi[i] == ii[j] || ii[i] == oi[j]) : oi[i] == oi[j])
            return fail(CN_ERR_ARG, "rotate_rows_many: a result would overwrite another rotation's operand or result");
    }
    if (deferring(ctx)) {                                   // queued like n cn_rotate_rows calls
        for (uint32_t i = 0; i < n; i++) CHECK(defer_staged(ctx, DOP_ROT, I, ii[i], nullptr, 0, nullptr, 0, O, oi[i], 1, steps[i]));
        return 0;
    }
    CHECK(cn_defer_flush(ctx));
    std::vector<RotJob> jobs(n);
    for (uint32_t i = 0; i < n; i++) jobs[i] = {I->d + (size_t)ii[i] * I->item_words, O->d + (size_t)oi[i] * O->item_words, steps[i], {}};
    return rotate_jobs(ctx, jobs);
API_END }
// out = acc + RotateRows(in, steps): the rotate-and-add step of SumAllSlots (AtomicSealBfvVector.cs:862-868) with the addition
// fused into the last kernel of the key switch.  Same words as cn_rotate_rows followed by cn_add.
static int rotate_rows_add_impl(cn_ctx *ctx, Buffer *I, uint32_t ii, int steps, Buffer *A, uint32_t ai, Buffer *O, uint32_t oi, uint32_t count) {
    if (!range_ok(I, ii, count) || !range_ok(A, ai, count) || !range_ok(O, oi, count)) return fail(CN_ERR_ARG, "index out of range");
    if (shifted_overlap(I, ii, O, oi, count) || shifted_overlap(A, ai, O, oi, count))          // (the fused accumulator is read where the result is stored: no path reads it ahead)
        return fail(CN_ERR_ARG, "rotate-and-add: operand / accumulator and result ranges overlap partially (use the same range or disjoint ranges)");
    if (!count) return 0;
    const uint64_t *i = I->d + ii * I->item_words, *a = A->d + ai * A->item_words; uint64_t *o = O->d + oi * O->item_words;
    if (steps == 0) {
        hipLaunchKernelGGL(k_addsub, dim3(count * 2 * ctx->hc.k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, i, a, o, ctx->dc, ctx->chunks, 0);
        HIPCHK(hipGetLastError()); launch_count(ctx);
        ctx->st.Addition += count;
        return 0;
    }
    if (has_direct_key(ctx, steps)) {
        CHECK(ensure_scratch(ctx, al(count * ctx->ctw2 * 8)));
        uint64_t *tmp = salloc<uint64_t>(ctx, count * ctx->ctw2);
        return do_galois(ctx, i, cn_galois_elt_from_step(ctx, steps), o, tmp, count, a);
    }
    // multi-hop (NAF) rotation: rotate into a staging array, then one element-wise add
    CHECK(ensure_scratch(ctx, al(count * ctx->ctw2 * 8) * 2));
    uint64_t *tmp = salloc<uint64_t>(ctx, count * ctx->ctw2), *stage = salloc<uint64_t>(ctx, count * ctx->ctw2);
    HIPCHK(hipMemcpyAsync(stage, i, count * ctx->ctw2 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    CHECK(rotate_rec(ctx, stage, steps, tmp, count));
    hipLaunchKernelGGL(k_addsub, dim3(count * 2 * ctx->hc.k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, a, stage, o, ctx->dc, ctx->chunks, 0);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    ctx->st.Addition += count;
    return 0;
}
static int rotate_columns_add_impl(cn_ctx *ctx, Buffer *I, uint32_t ii, Buffer *A, uint32_t ai, Buffer *O, uint32_t oi, uint32_t count) {
    if (!range_ok(I, ii, count) || !range_ok(A, ai, count) || !range_ok(O, oi, count)) return fail(CN_ERR_ARG, "index out of range");
    if (shifted_overlap(I, ii, O, oi, count) || shifted_overlap(A, ai, O, oi, count))          // (the fused accumulator is read where the result is stored: no path reads it ahead)
        return fail(CN_ERR_ARG, "rotate-and-add: operand / accumulator and result ranges overlap partially (use the same range or disjoint ranges)");
    if (!count) return 0;
    CHECK(ensure_scratch(ctx, al(count * ctx->ctw2 * 8)));
    uint64_t *tmp = salloc<uint64_t>(ctx, count * ctx->ctw2);
    return do_galois(ctx, I->d + ii * I->item_words, 2ull * ctx->hc.n - 1, O->d + oi * O->item_words, tmp, count, A->d + ai * A->item_words);
}
extern "C" int cn_rotate_rows_add(cn_ctx *ctx, cn_handle in, uint32_t ii, int steps, cn_handle acc, uint32_t ai, cn_handle out, uint32_t oi, uint32_t count) { API_BODY
    LOCK_ONLY; GETCT(I, in, 2); GETCT(A, acc, 2); GETCT(O, out, 2);
    if (deferring(ctx) && count && count <= DEFER_STAGED_MAX) {
        if (!range_ok(I, ii, count) || !range_ok(A, ai, count) || !range_ok(O, oi, count)) return fail(CN_ERR_ARG, "index out of range");
        CHECK(rotate_check(ctx, steps));
        return defer_staged(ctx, DOP_ROTADD, I, ii, A, ai, nullptr, 0, O, oi, count, steps);
    }
    CHECK(cn_defer_flush(ctx));
    return rotate_rows_add_impl(ctx, I, ii, steps, A, ai, O, oi, count);
API_END }
extern "C" int cn_rotate_columns_add(cn_ctx *ctx, cn_handle in, uint32_t ii, cn_handle acc, uint32_t ai, cn_handle out, uint32_t oi, uint32_t count) { API_BODY
    LOCK_ONLY; GETCT(I, in, 2); GETCT(A, acc, 2); GETCT(O, out, 2);
    if (deferring(ctx) && count && count <= DEFER_STAGED_MAX) {
        if (!range_ok(I, ii, count) || !range_ok(A, ai, count) || !range_ok(O, oi, count)) return fail(CN_ERR_ARG, "index out of range");
        if (!galois_key_present(ctx, 2ull * ctx->hc.n - 1)) return fail(CN_ERR_NOKEY, "Galois key not present");
        return defer_staged(ctx, DOP_COLSADD, I, ii, A, ai, nullptr, 0, O, oi, count, 0);
    }
    CHECK(cn_defer_flush(ctx));
    return rotate_columns_add_impl(ctx, I, ii, A, ai, O, oi, count);
API_END }
// SumAllSlots(length) of AtomicSealBfvVector.cs:888-935 on `count` single-block ciphertexts at once, in place: the column swap when
// length >= N/2, then log2 rotate-and-add steps (RotateRows(-2^s) + AddInplace).  length 0 = all N slots.
// the Galois elements of SumAllSlots(length) if the whole chain runs on the one-launch key switch with every link handing sigma_next(c1) on (N = 16384, batch); else empty
static std::vector<uint64_t> sum_slots_chain_elts(cn_ctx *ctx, uint32_t count, uint32_t length) {
    const uint32_t n = ctx->hc.n, half = n / 2;
    std::vector<uint64_t> elts;
    bool ok = ctx->ks_chain && count > 0;
    uint32_t l2 = length ? length : n;
    if (l2 >= half) { elts.push_back(2ull * n - 1); l2 = half; }
    for (uint32_t steps = 1; steps < l2 && ok; steps *= 2) { if (has_direct_key(ctx, -(int)steps)) elts.push_back(cn_galois_elt_from_step(ctx, -(int)steps)); else ok = false; }
    for (uint64_t e : elts) { auto it = ctx->gk.find(e); if (it == ctx->gk.end() || !it->second.d || !ks_pair14_ok(ctx, count, 1, it->second)) { ok = false; break; } }
    if (!ok || elts.size() < 2) elts.clear();
    return elts;
}
// first_ready: the scratch arena already holds sigma_(elts[0])(c1) of every ciphertext at its start (written by the producer of H: k_mul_plain_bcast) and is large enough
static int sum_slots_impl(cn_ctx *ctx, Buffer *H, uint32_t first, uint32_t count, uint32_t length, bool first_ready = false) {
    const uint32_t n = ctx->hc.n, half = n / 2;
    uint32_t len = length ? length : n;
    {   // N = 16384, batch: the links of the chain as ONE launch each - link s leaves sigma_(s+1) of its new c1 beside its result (k_keyswitch_pair14), so only the
        // first link needs a permutation pass (none when the producer has left it).  Same words as the loop below (the same key switches on the same operands).
        const std::vector<uint64_t> elts = sum_slots_chain_elts(ctx, count, length);
        if (elts.empty() && first_ready) return fail(CN_ERR_ARG, "internal: chained row-dot batch without a chain");
        if (!elts.empty()) {
            const size_t kn = (size_t)ctx->hc.k * n;
            if (first_ready) ctx->soff = 0; else CHECK(ensure_scratch(ctx, al(count * ctx->ctw2 * 8)));
            uint64_t *pp[2]; pp[0] = salloc<uint64_t>(ctx, count * ctx->ctw2); pp[1] = pp[0] + (size_t)count * kn;
            uint64_t *h = H->d + first * H->item_words;
            for (size_t s = 0; s < elts.size(); s++)
                CHECK(do_galois(ctx, h, elts[s], h, pp[s & 1], count, h, (s || first_ready) ? pp[s & 1] : nullptr, s + 1 < elts.size() ? elts[s + 1] : 0, pp[(s + 1) & 1]));
            return 0;
        }
    }
    if (len >= half) { CHECK(rotate_columns_add_impl(ctx, H, first, H, first, H, first, count)); len = half; }
    for (uint32_t steps = 1; steps < len; steps *= 2) CHECK(rotate_rows_add_impl(ctx, H, first, -(int)steps, H, first, H, first, count));
    return 0;
}
extern "C" int cn_sum_slots(cn_ctx *ctx, cn_handle h, uint32_t first, uint32_t count, uint32_t length) { API_BODY
    LOCK_ONLY; GETCT(H, h, 2);
    if (!range_ok(H, first, count)) return fail(CN_ERR_ARG, "index out of range");
    if (!count) return 0;
    if (deferring(ctx) && count <= DEFER_STAGED_MAX) {
        const uint32_t n = ctx->hc.n, half = n / 2;
        uint32_t len = length ? length : n;
        if (len >= half) { if (!galois_key_present(ctx, 2ull * n - 1)) return fail(CN_ERR_NOKEY, "Galois key not present"); len = half; }
        for (uint32_t st = 1; st < len; st *= 2) CHECK(rotate_check(ctx, -(int)st));
        return defer_staged(ctx, DOP_SUMSLOTS, H, first, nullptr, 0, nullptr, 0, H, first, count, length);
    }
    CHECK(cn_defer_flush(ctx));
    return sum_slots_impl(ctx, H, first, count, length);
API_END }
// out[r] = SumAllSlots(v * pt[r], length) for r < rows: every row of a plaintext matrix against ONE packed ciphertext
// (EncryptedSealBfvMatrix.Mul row-major, EncryptedSealBfvMatrix.cs:79-120 -> DotProduct, AtomicSealBfvVector.cs:963-977).
extern "C" int cn_rowdot_batch(cn_ctx *ctx, cn_handle v, uint32_t vi, cn_handle pt, uint32_t pi, uint32_t rows, uint32_t length, cn_handle out, uint32_t oi) { API_BODY
    LOCK; GETCT(V, v, 2); GETCT(O, out, 2); GETPT(P, pt);
    if (!rows) return 0;
    if (V == O && vi >= oi && vi < oi + rows) return fail(CN_ERR_ARG, "row-dot batch cannot overwrite its input");
    if (length != 1 && mul_plain_takes_bcast(ctx, rows)) {
        // the product kernel hands the chain its first permuted c1 (no k_galois_limbs pass over the products): one arena for the chain's two scratch arrays and the
        // transformed ciphertext, sized here and left where it is until the chain has run
        const std::vector<uint64_t> elts = sum_slots_chain_elts(ctx, rows, length);
        if (!elts.empty()) {
            CHECK(ensure_scratch(ctx, al(rows * ctx->ctw2 * 8) + al(V->item_words * 8)));
            uint64_t *p0 = salloc<uint64_t>(ctx, rows * ctx->ctw2), *ctn = salloc<uint64_t>(ctx, V->item_words);
            if (!p0 || !ctn) return fail(CN_ERR_HIP, "internal: scratch exhausted in the row-dot batch");
            const BcastNext nx{elts[0], p0, ctn};
            CHECK(mul_plain_impl(ctx, V, vi, true, P, pi, 1, O, oi, rows, &nx));
            return sum_slots_impl(ctx, O, oi, rows, length, true);
        }
    }
    CHECK(mul_plain_impl(ctx, V, vi, true, P, pi, 1, O, oi, rows));
    if (length == 1) return 0;
    return sum_slots_impl(ctx, O, oi, rows, length);
API_END }
extern "C" int cn_rotate_columns(cn_ctx *ctx, cn_handle in, uint32_t ii, cn_handle out, uint32_t oi, uint32_t count) { API_BODY
    LOCK_ONLY; GETCT(I, in, 2); GETCT(O, out, 2);
    if (deferring(ctx) && count && count <= DEFER_STAGED_MAX) {
        if (!range_ok(I, ii, count) || !range_ok(O, oi, count)) return fail(CN_ERR_ARG, "index out of range");
        if (!galois_key_present(ctx, 2ull * ctx->hc.n - 1)) return fail(CN_ERR_NOKEY, "Galois key not present");
        return defer_staged(ctx, DOP_COLS, I, ii, nullptr, 0, nullptr, 0, O, oi, count, 0);
    }
    CHECK(cn_defer_flush(ctx));
    return galois_impl(ctx, I, ii, 2ull * ctx->hc.n - 1, O, oi, count);
API_END }

// ---------------------------------------------------------------- client side on the device
static int set_plain_key(cn_ctx *ctx, uint64_t **slot, const uint64_t *words, size_t count, size_t expect, bool is_dev = false, bool coeff_form = false) {
    if (!words || count != expect) return fail(CN_ERR_ARG, "key has %zu words, expected %zu", count, expect);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (!*slot) HIPCHK(hipMalloc((void **)slot, expect * 8));
    HIPCHK(hipMemcpy(*slot, words, expect * 8, is_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    if (coeff_form) { CHECK(cn_run_ntt(ctx, *slot, (uint32_t)(expect / ctx->hc.n), 0, ctx->hc.k, 0)); HIPCHK(hipStreamSynchronize(ctx->stream)); }
    return 0;
}
// any key in either representation (include/cnhip.h)
extern "C" int cn_load_key(cn_ctx *ctx, int which, uint64_t elt, const uint64_t *words, size_t count, int is_dev, int form) { API_BODY
    LOCK; NOT_CAPTURING("cn_load_key");
    if (form != 0 && form != 1) return fail(CN_ERR_ARG, "key form must be 0 (NTT) or 1 (coefficients)");
    switch (which) {
        case 0: return set_key(ctx, ctx->rlk, words, count, cn_key_words(ctx, 0), is_dev, form == 1);
        case 1:
            if (!(elt & 1) || elt >= 2ull * ctx->hc.n) return fail(CN_ERR_ARG, "invalid Galois element");
            return set_key(ctx, ctx->gk[elt], words, count, cn_key_words(ctx, 1), is_dev, form == 1);
        case 2: return set_plain_key(ctx, &ctx->pk, words, count, ctx->ctw2, is_dev != 0, form == 1);
        case 3: return set_plain_key(ctx, &ctx->sk, words, count, ctx->ctw2 / 2, is_dev != 0, form == 1);
    }
    return fail(CN_ERR_ARG, "unknown key kind %d", which);
API_END }
extern "C" int cn_set_public_key(cn_ctx *ctx, const uint64_t *words, size_t count) { API_BODY LOCK; NOT_CAPTURING("cn_set_public_key"); return set_plain_key(ctx, &ctx->pk, words, count, ctx->ctw2); API_END }
extern "C" int cn_set_secret_key(cn_ctx *ctx, const uint64_t *words, size_t count) { API_BODY LOCK; NOT_CAPTURING("cn_set_secret_key"); return set_plain_key(ctx, &ctx->sk, words, count, ctx->ctw2 / 2); API_END }
// which: 0 relin, 1 galois(elt), 2 public, 3 secret.  Exports u64 residues (FP64-form keys are converted back).
extern "C" int cn_get_key(cn_ctx *ctx, int which, uint64_t elt, uint64_t *host, size_t count) { API_BODY
    LOCK; NOT_CAPTURING("cn_get_key");
    const uint64_t *src = nullptr; size_t words = 0; bool f64 = false;
    if (which == 0) { src = ctx->rlk.d; words = cn_key_words(ctx, 0); f64 = ctx->rlk.f64; }
    else if (which == 1) { auto it = ctx->gk.find(elt); if (it != ctx->gk.end()) { src = it->second.d; f64 = it->second.f64; } words = cn_key_words(ctx, 1); }
    else if (which == 2) { src = ctx->pk; words = ctx->ctw2; }
    else if (which == 3) { src = ctx->sk; words = ctx->ctw2 / 2; }
    if (!src) return fail(CN_ERR_NOKEY, "key not present");
    if (!host || count != words) return fail(CN_ERR_ARG, "key has %zu words", words);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(host, src, words * 8, hipMemcpyDeviceToHost));
    if (f64) for (size_t i = 0; i < words; i++) { double d; memcpy(&d, &host[i], 8); host[i] = (uint64_t)d; }
    return 0;
API_END }
static RngKey rng_key_of(const cn_ctx *ctx) { RngKey k; memcpy(k.k, ctx->rng_key, sizeof k.k); return k; }
// `polys` polynomials [polys][k][N] of residues: kind 0 ternary, 1 clipped normal (both drawn ONCE per coefficient into an int8 array in
// scratch - the caller's ensure_scratch leaves room for polys * N bytes - and expanded to the k limbs), 2 uniform per limb
static int sample_poly(cn_ctx *ctx, uint64_t *dst, uint32_t polys, int kind, uint64_t seed, uint64_t stream) {
    const uint32_t n = ctx->hc.n, k = ctx->hc.k;
    if (n < 16) return fail(CN_ERR_ARG, "device sampling needs N >= 16");
    if (kind == 2) {
        const uint64_t threads = (uint64_t)polys * k * (n / 8);
        hipLaunchKernelGGL(k_sample_uniform, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, dst, ctx->dc, polys, rng_key_of(ctx), seed, (uint32_t)stream, ctx->rng_item);
    } else {
        int8_t *small = salloc<int8_t>(ctx, (size_t)polys * n);
        if (!small) return fail(CN_ERR_HIP, "internal: scratch exhausted in the sampler");
        const uint64_t threads = (uint64_t)polys * (n / (kind == 0 ? 16 : 8));
        hipLaunchKernelGGL(k_sample_small, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, small, n, kind, 1u, polys, rng_key_of(ctx), seed, (uint32_t)stream,
                           ctx->rng_item, (const EncTab *)nullptr);
        hipLaunchKernelGGL(k_expand_small, dim3(polys * k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, small, dst, ctx->dc, ctx->chunks);
        launch_count(ctx);
    }
    HIPCHK(hipGetLastError()); launch_count(ctx);
    ctx->rng_item += polys;
    return 0;
}
// one key-switch key for the NTT-form target polynomial snew: [(l,d)][2][k][N]
static int gen_ksk(cn_ctx *ctx, const uint64_t *snew, int dbc, const uint32_t *dig, uint32_t tot, uint64_t seed, uint64_t *key, uint64_t *e) {
    const uint32_t n = ctx->hc.n, k = ctx->hc.k; const size_t kn = (size_t)k * n;
    uint64_t *p = key;
    for (uint32_t l = 0; l < k; l++) {
        for (uint32_t d = 0; d < dig[l]; d++, p += 2 * kn) {
            CHECK(sample_poly(ctx, p + kn, 1, 2, seed, 3));                 // a: uniform, directly in the NTT domain
            CHECK(sample_poly(ctx, e, 1, 1, seed, 1));
            CHECK(cn_run_ntt(ctx, e, k, 0, k, 0));
            // message term 2^(dbc d) snew in limb l only; "ks_xi": the RNS image of (q/q_l) 2^(dbc d) snew = (q/q_l mod q_l) 2^(dbc d) snew in limb l, zero elsewhere (DevConsts::ks_xi)
            KeyFactors fac{};
            for (uint32_t j = 0; j < k; j++) {
                if (!ctx->hc.ks_xi && j != l) continue;
                const uint64_t qj = ctx->hc.q[j].q; unsigned __int128 f = 1;
                for (uint32_t i = 0; i < d; i++) f = (f << dbc) % qj;
                if (ctx->hc.ks_xi) f = f * ctx->hc.qhat_q[l][j] % qj;
                fac.f[j] = (uint64_t)f;
            }
            hipLaunchKernelGGL(k_key_b, dim3(k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, p + kn, e, ctx->sk, snew, fac, p, ctx->dc, ctx->chunks);
            HIPCHK(hipGetLastError()); launch_count(ctx);
        }
    }
    (void)tot;
    return 0;
}
static int adopt_ksk(cn_ctx *ctx, KsKey &slot, uint64_t *dev, size_t words) {          // takes ownership of a device buffer
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (slot.owned && slot.d) HIPCHK(hipFree(slot.d));
    slot = {dev, true, false};
    if (keys_as_f64(ctx)) {
        hipLaunchKernelGGL(k_u64_to_f64, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctx->stream, dev, words);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        slot.f64 = true;
    }
    return 0;
}
// sampler key material: the 256-bit ChaCha20 key of every block keygen / encrypt draw from now on (cn_set_rng_salt: its first 64 bits)
extern "C" int cn_set_rng_salt(cn_ctx *ctx, uint64_t salt) { API_BODY LOCK; ctx->rng_key[0] = (uint32_t)salt; ctx->rng_key[1] = (uint32_t)(salt >> 32); return 0; API_END }
// known-answer hook: the generator's block for (key, counter words 12-13, nonce words 14-15) - RFC 7539 section 2.3.2 is reproduced with
// counter = 0x09000000'00000001, nonce = 0x00000000'4a000000 (tests/test_gpu_client.py)
extern "C" int cn_rng_selftest(cn_ctx *ctx, const uint8_t *key32, uint64_t counter, uint64_t nonce, uint32_t *out16) { API_BODY
    LOCK; NOT_CAPTURING("cn_rng_selftest");
    if (!key32 || !out16) return fail(CN_ERR_ARG, "null argument");
    RngKey k; memcpy(k.k, key32, 32);
    CHECK(ensure_scratch(ctx, 256));
    uint32_t *d = salloc<uint32_t>(ctx, 16);
    hipLaunchKernelGGL(k_rng_block, dim3(1), dim3(1), 0, ctx->stream, k, counter, nonce, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out16, d, 64, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
API_END }
extern "C" int cn_set_rng_key(cn_ctx *ctx, const uint8_t *key32) { API_BODY
    LOCK;
    if (!key32) return fail(CN_ERR_ARG, "null argument");
    memcpy(ctx->rng_key, key32, 32);
    return 0;
API_END }
// KeyGenerator (AtomicSealBfvVector.cs:62-74,163-173 runs it inside SEAL): secret, public, relinearisation and the default Galois
// key set (2N-1, 3^(2^i), 3^(-2^i)) generated on the device from the ChaCha20 sampler.
extern "C" int cn_keygen(cn_ctx *ctx, uint64_t seed, int with_galois) { API_BODY
    LOCK; NOT_CAPTURING("cn_keygen");
    const uint32_t n = ctx->hc.n, k = ctx->hc.k; const size_t kn = (size_t)k * n;
    if (!ctx->sk) HIPCHK(hipMalloc((void **)&ctx->sk, kn * 8));
    if (!ctx->pk) HIPCHK(hipMalloc((void **)&ctx->pk, 2 * kn * 8));
    // the sampler carves an N-byte int8 array out of the scratch arena per call and keygen makes ~2 such calls per key digit: room for all of them
    const size_t draws = 4 + 2 * ((size_t)ctx->hc.rl_tot + (with_galois ? (size_t)ctx->hc.gk_tot * (2 * ctx->hc.logn) : 0));
    CHECK(ensure_scratch(ctx, al(kn * 8) * 4 + draws * al(n)));
    uint64_t *e = salloc<uint64_t>(ctx, kn), *snew = salloc<uint64_t>(ctx, kn), *tmp = salloc<uint64_t>(ctx, kn);
    ctx->rng_item = 0;
    CHECK(sample_poly(ctx, ctx->sk, 1, 0, seed, 0));
    CHECK(cn_run_ntt(ctx, ctx->sk, k, 0, k, 0));
    // public key (-(a s + e), a)
    CHECK(sample_poly(ctx, ctx->pk + kn, 1, 2, seed, 3));
    CHECK(sample_poly(ctx, e, 1, 1, seed, 1));
    CHECK(cn_run_ntt(ctx, e, k, 0, k, 0));
    hipLaunchKernelGGL(k_key_b, dim3(k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, ctx->pk + kn, e, ctx->sk, ctx->sk, KeyFactors{}, ctx->pk, ctx->dc, ctx->chunks);
    HIPCHK(hipGetLastError());
    // relinearisation key: target s^2
    hipLaunchKernelGGL(k_mul_limbs, dim3(k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, ctx->sk, ctx->sk, snew, ctx->dc, ctx->chunks);
    HIPCHK(hipGetLastError());
    uint64_t *rl; size_t rlw = cn_key_words(ctx, 0);
    HIPCHK(hipMalloc((void **)&rl, rlw * 8));
    CHECK(gen_ksk(ctx, snew, ctx->hc.dbc, ctx->hc.rl_dig, ctx->hc.rl_tot, seed, rl, e));
    CHECK(adopt_ksk(ctx, ctx->rlk, rl, rlw));
    if (with_galois) {
        const uint64_t m = 2ull * n; std::vector<uint64_t> elts{m - 1};
        uint64_t p3 = 3, ip3 = 0;
        for (uint64_t x = 1; x < m; x += 2) if (((x * 3) & (m - 1)) == 1) { ip3 = x; break; }
        for (uint32_t i = 0; i + 1 < ctx->hc.logn; i++) { elts.push_back(p3); p3 = (p3 * p3) & (m - 1); elts.push_back(ip3); ip3 = (ip3 * ip3) & (m - 1); }
        size_t gw = cn_key_words(ctx, 1);
        for (uint64_t elt : elts) {
            HIPCHK(hipMemcpyAsync(tmp, ctx->sk, kn * 8, hipMemcpyDeviceToDevice, ctx->stream));
            CHECK(cn_run_ntt(ctx, tmp, k, 0, k, 1));
            hipLaunchKernelGGL(k_galois, dim3(k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, tmp, snew, ctx->dc, ctx->chunks, elt);
            HIPCHK(hipGetLastError());
            CHECK(cn_run_ntt(ctx, snew, k, 0, k, 0));
            uint64_t *gk; HIPCHK(hipMalloc((void **)&gk, gw * 8));
            CHECK(gen_ksk(ctx, snew, ctx->hc.gdbc, ctx->hc.gk_dig, ctx->hc.gk_tot, seed, gk, e));
            CHECK(adopt_ksk(ctx, ctx->gk[elt], gk, gw));
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
API_END }
// Encryptor.Encrypt (AtomicSealBfvVector.cs:1211,1227): (pk0 u + e1 + Delta m [+ r_t(q)], pk1 u + e2); pt = 0 encrypts zero.
// tab != null: `cnt` encryptions whose outputs / plaintexts / nonces / items come from the table (host copy `htab`), else dense out / ptd
static int encrypt_chain(cn_ctx *ctx, uint32_t cnt, const uint64_t *ptd, uint32_t pt_stride_words, uint64_t *out, uint64_t seed, const EncTab *htab) {
    const uint32_t n = ctx->hc.n, k = ctx->hc.k; const size_t kn = (size_t)k * n;
    CHECK(ensure_scratch(ctx, al((size_t)cnt * kn * 8) + al((size_t)cnt * n) + al((size_t)cnt * 2 * n) + (htab ? al(cnt * sizeof(EncTab)) : 0) + 1024));
    uint64_t *u = salloc<uint64_t>(ctx, (size_t)cnt * kn);
    int8_t *us = salloc<int8_t>(ctx, (size_t)cnt * n), *es = salloc<int8_t>(ctx, (size_t)cnt * 2 * n);
    EncTab *dtab = nullptr;
    if (htab) CHECK(upload_tmp(ctx, htab, cnt, &dtab));
    if (!u || !us || !es) return fail(CN_ERR_HIP, "internal: scratch exhausted in encrypt");
    const RngKey key = rng_key_of(ctx);
    const uint64_t item0 = ctx->rng_item;
    hipLaunchKernelGGL(k_sample_small, dim3((unsigned)(((uint64_t)cnt * (n / 16) + 255) / 256)), dim3(256), 0, ctx->stream, us, n, 0, 1u, cnt, key, seed, 0u, item0, (const EncTab *)dtab);
    hipLaunchKernelGGL(k_sample_small, dim3((unsigned)(((uint64_t)cnt * 2 * (n / 8) + 255) / 256)), dim3(256), 0, ctx->stream, es, n, 1, 2u, cnt, key, seed, 1u, item0, (const EncTab *)dtab);
    if (!htab) ctx->rng_item += cnt;
    const bool f64 = ctx->use_f64 && ctx->hc.q_f64;
    if (ctx->enc_fused && !ctx->legacy_ntt) {                 // one kernel behind the samplers: u stays in registers between its transform and the two components
        uint64_t qmax = 0; for (uint32_t j = 0; j < k; j++) qmax = std::max(qmax, ctx->hc.q[j].q);
        const int pol = f64 ? ((qmax >> 44) ? POL_F64 : POL_F64L) : POL_U64;
        if (rr_ops[pol]->enc_fused(ctx, us, ptd, pt_stride_words, out, cnt, es, dtab)) {
            HIPCHK(hipGetLastError()); launch_count(ctx, 3);
            ctx->st.ntt_forward_limbs += (uint64_t)cnt * k;          // (counted like the three-launch chain)
            return 0;
        }
    }
    hipLaunchKernelGGL(k_expand_small, dim3(cnt * k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, us, u, ctx->dc, ctx->chunks);
    HIPCHK(hipGetLastError()); launch_count(ctx, 3);
    CHECK(cn_run_ntt(ctx, u, cnt * k, 0, k, 0));
    if (!rr_ops[f64 ? POL_F64 : POL_U64]->enc_tail(ctx, u, ptd, pt_stride_words, out, cnt, es, dtab)) return fail(CN_ERR_ARG, "unsupported size");
    HIPCHK(hipGetLastError()); launch_count(ctx);
    return 0;
}
static int defer_encrypt(cn_ctx *ctx, const uint64_t *ptd, uint32_t pt_stride_words, Buffer *O, uint32_t oi, uint32_t count, uint64_t seed);
static int encrypt_body(cn_ctx *ctx, cn_handle pt, uint32_t pi, uint32_t pt_stride, cn_handle out, uint32_t oi, uint32_t count, uint64_t seed);
extern "C" int cn_encrypt(cn_ctx *ctx, cn_handle pt, uint32_t pi, uint32_t pt_stride, cn_handle out, uint32_t oi, uint32_t count, uint64_t seed) {
    if (submit_async(ctx) && count && count <= 4) return ring_push(ctx, SUB_ENCRYPT, count, 0, 0, pt, pi, out, oi, pt_stride, seed);
    API_BODY LOCK_ONLY; return encrypt_body(ctx, pt, pi, pt_stride, out, oi, count, seed); API_END
}
static int encrypt_body(cn_ctx *ctx, cn_handle pt, uint32_t pi, uint32_t pt_stride, cn_handle out, uint32_t oi, uint32_t count, uint64_t seed) {
    NOT_CAPTURING("cn_encrypt (a replayed graph would reuse its randomness)"); GETCT(O, out, 2);
    if (!ctx->pk) return fail(CN_ERR_NOKEY, "public key not set");
    if (!range_ok(O, oi, count)) return fail(CN_ERR_ARG, "index out of range");
    if (ctx->hc.logn < 10 || ctx->hc.logn > 14) return fail(CN_ERR_ARG, "device encryption needs 1024 <= N <= 16384");
    const uint64_t *ptd = nullptr;
    if (pt) { Buffer *P = getbuf(ctx, pt, 1); if (!P || !range_ok(P, pi, pt_stride ? count : 1, pt_stride ? pt_stride : 1)) return fail(CN_ERR_ARG, "invalid plaintext range"); ptd = P->d + (size_t)pi * ctx->hc.n; }
    if (!count) return 0;
    // per-ciphertext callers (PoolLayer.ElementAt encrypts a zero vector per padded tap, PoolLayer.cs:67-80): queued like the evaluator calls
    if (deferring(ctx) && count <= 4) return defer_encrypt(ctx, ptd, pt_stride ? ctx->hc.n : 0, O, oi, count, seed);
    CHECK(cn_defer_flush(ctx));
    return encrypt_chain(ctx, count, ptd, pt_stride ? ctx->hc.n : 0, O->d + oi * O->item_words, seed, nullptr);
}
// AllocateCiphertext + Encryptor.Encrypt(PlainZero) in ONE call (the unchanged PoolLayer does both per padded convolution tap, PoolLayer.cs:67-80,
// AtomicSealBfvVector.cs:566): one lock acquisition instead of two, same queue entry / same words as cn_ct_alloc followed by cn_encrypt(pt = 0)
extern "C" int cn_encrypt_zero_new(cn_ctx *ctx, uint64_t seed, cn_handle *out) {
    if (out && submit_async(ctx)) {                     // a ready handle + one record (same queue entry as the locked path below)
        const cn_handle h = ctx->ready->pop();
        if (h) { *out = h; return ring_push(ctx, SUB_ENCRYPT_ZERO, 1, 0, 0, 0, 0, h, 0, 0, seed); }
    }
    API_BODY
    LOCK_ONLY; NOT_CAPTURING("cn_encrypt_zero_new (a replayed graph would reuse its randomness)");
    if (!out) return fail(CN_ERR_ARG, "null argument");
    if (!ctx->pk) return fail(CN_ERR_NOKEY, "public key not set");
    if (ctx->hc.logn < 10 || ctx->hc.logn > 14) return fail(CN_ERR_ARG, "device encryption needs 1024 <= N <= 16384");
    cn_handle h = 0;
    CHECK(alloc_buf(ctx, 0, 1, 2, &h));
    if (ctx->defer.load(std::memory_order_relaxed) == 2) ready_refill(ctx);          // (the ring of ready handles had run dry)
    Buffer *O = ctx->bufs.find(h);
    int rc;
    if (deferring(ctx)) rc = defer_encrypt(ctx, nullptr, 0, O, 0, 1, seed);
    else { rc = cn_defer_flush(ctx); if (!rc) rc = encrypt_chain(ctx, 1, nullptr, 0, O->d, seed, nullptr); }
    if (rc) { (void)dev_release(ctx, O->d, O->item_words * 8); ctx->bufs.erase(h); return rc; }
    *out = h;
    return 0;
API_END }
#define DISPATCH_K2(fn, ...) switch (ctx->hc.k) { \
    case 1: fn<1>(__VA_ARGS__); break; case 2: fn<2>(__VA_ARGS__); break; case 3: fn<3>(__VA_ARGS__); break; case 4: fn<4>(__VA_ARGS__); break; \
    case 5: fn<5>(__VA_ARGS__); break; case 6: fn<6>(__VA_ARGS__); break; case 7: fn<7>(__VA_ARGS__); break; case 8: fn<8>(__VA_ARGS__); break; \
    case 9: fn<9>(__VA_ARGS__); break; default: return fail(CN_ERR_ARG, "at most 9 coefficient moduli"); }
template <int K> static void launch_dec_scale(cn_ctx *c, const uint64_t *c0, size_t stride, const uint64_t *acc, uint64_t *plain, uint32_t cnt) {
    hipLaunchKernelGGL(k_decrypt_scale<K>, dim3(cnt * c->chunks), dim3(c->bs), 0, c->stream, c0, stride, acc, plain, c->dc, c->chunks);
}
// acc[ct][j] <- c1 s (+ c2 s^2) in coefficient form: the part of the decryption phase that needs the secret key
static int decrypt_phase(cn_ctx *ctx, Buffer *I, uint32_t ci, uint32_t count, uint64_t *&acc) {
    const uint32_t n = ctx->hc.n, k = ctx->hc.k; const size_t kn = (size_t)k * n;
    CHECK(ensure_scratch(ctx, al((size_t)count * kn * 8) * 3 + al(kn * 8)));
    acc = salloc<uint64_t>(ctx, (size_t)count * kn);
    uint64_t *tmp = salloc<uint64_t>(ctx, (size_t)count * kn), *sp = salloc<uint64_t>(ctx, kn);
    const uint64_t *base = I->d + ci * I->item_words;
    HIPCHK(hipMemcpy2DAsync(acc, kn * 8, base + kn, I->item_words * 8, kn * 8, count, hipMemcpyDeviceToDevice, ctx->stream));
    CHECK(cn_run_ntt(ctx, acc, count * k, 0, k, 0));
    hipLaunchKernelGGL(k_mul_limbs_bcast, dim3(count * k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, acc, ctx->sk, (const uint64_t *)nullptr, acc, ctx->dc, ctx->chunks);
    if (I->size == 3) {
        hipLaunchKernelGGL(k_mul_limbs, dim3(k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, ctx->sk, ctx->sk, sp, ctx->dc, ctx->chunks);
        HIPCHK(hipMemcpy2DAsync(tmp, kn * 8, base + 2 * kn, I->item_words * 8, kn * 8, count, hipMemcpyDeviceToDevice, ctx->stream));
        CHECK(cn_run_ntt(ctx, tmp, count * k, 0, k, 0));
        hipLaunchKernelGGL(k_mul_limbs_bcast, dim3(count * k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, tmp, sp, acc, acc, ctx->dc, ctx->chunks);
    }
    HIPCHK(hipGetLastError()); launch_count(ctx, 2);
    return cn_run_ntt(ctx, acc, count * k, 0, k, 1);
}
// Decryptor.Decrypt (AtomicSealBfvVector.cs:1042,1085): m = round(t (c0 + c1 s + c2 s^2) / q) mod t
extern "C" int cn_decrypt(cn_ctx *ctx, cn_handle ct, uint32_t ci, uint32_t count, cn_handle pt_out, uint32_t pi) { API_BODY
    LOCK; GETCT(I, ct, 0); GETPT(P, pt_out);
    if (!ctx->sk) return fail(CN_ERR_NOKEY, "secret key not set");
    if (!ctx->hc.inv_g_t) return fail(CN_ERR_ARG, "device decryption needs a prime plain modulus");
    if (!range_ok(I, ci, count) || !range_ok(P, pi, count)) return fail(CN_ERR_ARG, "index out of range");
    if (!count) return 0;
    uint64_t *acc = nullptr;
    CHECK(decrypt_phase(ctx, I, ci, count, acc));
    DISPATCH_K2(launch_dec_scale, ctx, I->d + ci * I->item_words, I->item_words, acc, P->d + (size_t)pi * ctx->hc.n, count);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    for (uint32_t c = 0; c < count; c++) P->pt_zero[pi + c] = 0;      // unknown: treated as non-zero
    return 0;
API_END }
// Decryptor.InvariantNoiseBudget (CryptoTracker.cs:41-52): the residues of t (c0 + c1 s + c2 s^2) mod q, [count][k][N] to the host
extern "C" int cn_noise_poly(cn_ctx *ctx, cn_handle ct, uint32_t ci, uint32_t count, uint64_t *host) { API_BODY
    LOCK; NOT_CAPTURING("cn_noise_poly"); GETCT(I, ct, 0);
    if (!ctx->sk) return fail(CN_ERR_NOKEY, "secret key not set");
    if (!host || !range_ok(I, ci, count)) return fail(CN_ERR_ARG, "index out of range");
    if (!count) return 0;
    uint64_t *acc = nullptr;
    CHECK(decrypt_phase(ctx, I, ci, count, acc));
    hipLaunchKernelGGL(k_noise_poly, dim3(count * ctx->hc.k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, I->d + ci * I->item_words, (size_t)I->item_words, acc, ctx->dc, ctx->chunks);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    HIPCHK(hipMemcpyAsync(host, acc, (size_t)count * ctx->hc.k * ctx->hc.n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
API_END }


// ---------------------------------------------------------------- deferred submission of per-ciphertext calls
// The reference's layers call the evaluator one ciphertext at a time from Defaults.ThreadCount threads: PoolLayer.Apply issues one
// DenseMatrixBySparseVectorMultiply + Add per (map, corner) (NeuralNetworks/PoolLayer.cs:113-121,182,214), ElementWiseMultiply one
// Multiply + Relinearize per column (HE Wrapper/EncryptedSealBfvMatrix.cs:140-154), each through Utils.ParallelProcessInEnv
// (HE Wrapper/Utils.cs:46-88).  With cn_set_option("defer", 1) such calls are not launched one by one: they are QUEUED with the device
// addresses of their operands, ordered by a dependency level (read-after-write, write-after-read and write-after-write hazards on whole
// ciphertexts), and flushed as a handful of batched launches - all pending calls of one level and one kind become ONE launch of the
// same kernels the batched entry points use, reading their operands through address tables.  A flush happens when a call arrives
// whose level is deeper than anything queued (the previous layer is then complete: its callers had to wait for it), when the queue is
// full, and before every entry point that is not deferrable (cn_sync, downloads, rotations, key changes ...).  Results are the same
// words as the immediate calls - every operation is exact modular arithmetic, batching changes no value.  Errors of a flush (HIP
// failures) surface at the call that triggered it; argument errors are still reported by the call that made them.
// A flush is triggered by demand (any entry point that needs results), by a full queue, and at LAYER BOUNDARIES, so that the device works on
// one layer while the callers queue the next.  A boundary is recognised by the "heavy depth" of a value: 0 for anything that was not
// produced by a queued call, and for fresh encryptions; a scalar product (DenseMatrixBySparseVectorMultiply) or a Multiply + Relinearize
// produces depth 1 + the deepest of its inputs; additions, plaintext products, rotations and copies pass the depth of their inputs on.  A heavy call that would reach depth 2 reads
// the result of another queued heavy call: the layer that produced it is complete (its callers have returned) - everything queued is
// launched, if at least DEFER_FLUSH_MIN calls wait.  (Round 2 used the plain dependency level for this; with the literal padded taps -
// encryption -> scalar product -> plain addition inside ONE layer - several caller threads interleave those levels and the layer was cut
// into dozens of small launches: 0.47 of the batched rate at 4-32 threads against 0.84 at one.  Flushing whenever the stream had run dry
// instead cut the first layers into 256-call pieces and lost the bias folding: 0.85 against 0.93.)
static const size_t DEFER_FLUSH_MIN = 64, DEFER_MAX_OPS = 32768;
static DeferQueue *cn_defer_new() { return new DeferQueue(); }
static void cn_defer_delete(DeferQueue *q) { delete q; }
static bool cn_defer_pending(cn_ctx *ctx) { return ctx->dq && !ctx->dq->ops.empty(); }

static int32_t defer_level(DeferQueue *q, const uint64_t *const *ins, uint32_t nin, const uint64_t *out) {
    int32_t lv = 0;
    for (uint32_t i = 0; i < nin; i++) {
        if (!ins[i]) continue;
        const DeferQueue::Haz *h = q->haz.find(ins[i]);
        if (h) lv = std::max(lv, h->w + 1);
    }
    const DeferQueue::Haz *ho = q->haz.find(out);
    if (ho) lv = std::max(lv, std::max(ho->w, ho->r) + 1);
    return lv;
}
// queue one operation; ins: the ciphertexts it reads
static int defer_push(cn_ctx *ctx, DOp op, const uint64_t *const *ins, uint32_t nin) {
    DeferQueue *q = ctx->dq;
    int32_t lv = defer_level(q, ins, nin, op.out);
    const bool heavy = op.type == DOP_GEMM1 || op.type == DOP_MULRELIN;      // (rotations counted as heavy too was measured: the LoLa rows were cut into more, smaller
                                                                               // launches - 388 instead of 309 per prime, 14.3 instead of 12.1 ms per image)
    int32_t hd = 0;
    for (uint32_t i = 0; i < nin; i++) if (ins[i]) { const DeferQueue::Haz *h = q->haz.find(ins[i]); if (h) hd = std::max(hd, h->hd); }
    hd += heavy ? 1 : 0;
    if (q->ops.size() >= DEFER_MAX_OPS || (heavy && hd >= 2 && q->ops.size() >= DEFER_FLUSH_MIN)) {
        // a layer boundary (or a full queue): launch what is queued, the callers go on queueing the next layer behind it
        std::vector<uint64_t> ta, tw;
        if (op.type == DOP_GEMM1) {                     // the terms of this call sit at the end of the term arrays: keep them over the flush
            ta.assign(q->addr.begin() + op.terms, q->addr.end()); tw.assign(q->wt.begin() + op.terms, q->wt.end());
            q->addr.resize(op.terms); q->wt.resize(op.terms);
        }
        CHECK(cn_defer_flush(ctx));
        if (op.type == DOP_GEMM1) { op.terms = 0; q->addr = ta; q->wt = tw; }
        lv = 0; hd = heavy ? 1 : 0;
    }
    op.level = lv;
    const int32_t me = (int32_t)q->ops.size();
    for (uint32_t i = 0; i < nin; i++) {
        if (!ins[i]) continue;
        DeferQueue::Haz &h = q->haz[ins[i]];
        h.r = std::max(h.r, lv); h.readers++;
    }
    DeferQueue::Haz &ho = q->haz[op.out];
    ho.w = lv; ho.r = -1; ho.wop = me; ho.readers = 0; ho.hd = hd;
    q->maxlevel = std::max(q->maxlevel, lv);
    q->ops.push_back(op);
    return 0;
}

// element-wise kernels over address tables: entry c = {a, b, out} (b: second ciphertext or plaintext polynomial)
struct Tab3 { const NTT_GLOBAL uint64_t *a, *b; NTT_GLOBAL uint64_t *out; };          // global addresses (global_load / global_store, not flat)
__global__ void k_addsub_tab(const Tab3 *__restrict__ tab, const DevConsts *__restrict__ C, uint32_t chunks, int op) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t per = 2 * C->k, ct = limb / per, l = limb % per;
    const Tab3 t = tab[ct];
    const uint64_t q = C->q[l % C->k].q; const size_t o = (size_t)l * C->n + i;
    const uint64_t x = t.a[o];
    t.out[o] = op == 0 ? addmod(x, t.b[o], q) : submod(x, t.b[o], q);
}
__global__ void k_add_plain_tab(const Tab3 *__restrict__ tab, const DevConsts *__restrict__ C, uint32_t chunks, int subtract) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t k = C->k, per = 2 * k, ct = limb / per, l = limb % per, j = l % k;
    const Tab3 t = tab[ct];
    const size_t o = (size_t)l * C->n + i;
    uint64_t x = t.a[o];
    if (l < k) {
        const uint64_t s = scale_plain(C, t.b[i], j), q = C->q[j].q;
        x = subtract ? submod(x, s, q) : addmod(x, s, q);
    }
    t.out[o] = x;
}

// all queued DenseMatrixBySparseVectorMultiply calls of one level with K terms each: ONE scalar GEMM over address tables
static int flush_gemm_group(cn_ctx *ctx, DeferQueue *q, const std::vector<const DOp *> &ops, uint32_t K) {
    // group the outputs that gather the same inputs (PoolLayer: every map of one corner shares its patch; a dense layer: one group)
    std::map<std::vector<uint64_t>, std::vector<const DOp *>> groups;
    for (const DOp *op : ops) groups[std::vector<uint64_t>(q->addr.begin() + op->terms, q->addr.begin() + op->terms + K)].push_back(op);
    const uint32_t G = (uint32_t)groups.size();
    uint32_t M = 0;
    for (auto &g : groups) M = std::max<uint32_t>(M, (uint32_t)g.second.size());
    bool wsmall = true;
    for (const DOp *op : ops) wsmall = wsmall && gemm_weights_small(ctx, &q->wt[op->terms], K);
    const GemmArith ar = gemm_arith(ctx, wsmall);
    const bool mfma = gemm_mfma_ok(ctx, ar, M, K);
    const uint32_t Kp = mfma ? ((K + 31) / 32) * 32 : ((K + 15) & ~15u) + 16;       // gather rows: 16 spare entries (the VALU kernels request up to 2 x 8 terms ahead)
    std::vector<uint64_t> hidx((size_t)G * Kp, 0), hoidx((size_t)G * M, 0), hbidx((size_t)G * M, 0);
    std::vector<const DOp *> member((size_t)G * M, nullptr);
    bool any_bias = false;
    const uint64_t *fallback = nullptr;
    {
        uint32_t g = 0;
        for (auto &kv : groups) {
            memcpy(&hidx[(size_t)g * Kp], kv.first.data(), (size_t)K * 8);
            for (uint32_t m = 0; m < kv.second.size(); m++) {
                const DOp *op = kv.second[m];
                member[(size_t)g * M + m] = op; hoidx[(size_t)g * M + m] = (uint64_t)op->out; hbidx[(size_t)g * M + m] = (uint64_t)op->bias;
                any_bias = any_bias || op->bias;
            }
            for (uint64_t a : kv.first) if (a && !fallback) fallback = (const uint64_t *)a;
            g++;
        }
    }
    const bool small = ar.small, two = ar.two; const uint32_t lazy = ar.lazy; uint32_t MT = 1, WP = 0;
    bool one = false;
    std::vector<char> wbytes;
    auto row = [&](uint32_t g, uint32_t m) -> const uint64_t * { const DOp *op = member[(size_t)g * M + m]; return op ? &q->wt[op->terms] : nullptr; };
    if (mfma) {
        for (const DOp *op : ops) WP = std::max(WP, gemm_weight_planes(ctx, &q->wt[op->terms], K));
        pack_gemm_mfma(ctx, G, M, K, WP, row, [&](uint32_t g, uint32_t kk) { return hidx[(size_t)g * Kp + kk] != 0; }, wbytes);
    } else {
        auto tap = [&](uint32_t g, uint32_t kk) { return hidx[(size_t)g * Kp + kk] != 0; };
        pack_gemm_weights(ctx, G, M, K, small, row, tap, MT, wbytes);
        one = gemm_one_limb(ctx, ar, G, M, K, row, tap);
    }
    const size_t off_oidx = al(hidx.size() * 8), off_bidx = off_oidx + al(hoidx.size() * 8), off_w = off_bidx + al(hbidx.size() * 8);
    std::vector<char> host(off_w + al(wbytes.size()), 0);
    memcpy(host.data(), hidx.data(), hidx.size() * 8);
    memcpy(host.data() + off_oidx, hoidx.data(), hoidx.size() * 8);
    memcpy(host.data() + off_bidx, hbidx.data(), hbidx.size() * 8);
    memcpy(host.data() + off_w, wbytes.data(), wbytes.size());
    CHECK(ensure_scratch(ctx, al(host.size())));
    char *tables; CHECK(upload_tmp(ctx, host.data(), host.size(), &tables));
    GemmLaunch gl{small, two, true, MT, fallback, tables, tables + off_w, tables + off_oidx, nullptr, any_bias ? tables + off_bidx : nullptr, nullptr,
                  G, M, K, lazy, Kp, 0, WP, (M + 31) / 32, (K + 31) / 32, 2, (uint32_t)ctx->gemm_order, one};
    return mfma ? cn_l_gemm_mfma(ctx, gl) : cn_l_gemm(ctx, gl);
}
static int flush_elementwise_group(cn_ctx *ctx, const std::vector<const DOp *> &ops, int type) {
    std::vector<Tab3> tab(ops.size());
    for (size_t i = 0; i < ops.size(); i++) tab[i] = {(const NTT_GLOBAL uint64_t *)ops[i]->a, (const NTT_GLOBAL uint64_t *)ops[i]->b, (NTT_GLOBAL uint64_t *)ops[i]->out};
    CHECK(ensure_scratch(ctx, al(tab.size() * sizeof(Tab3))));
    Tab3 *dt; CHECK(upload_tmp(ctx, tab.data(), tab.size(), &dt));
    const uint32_t limbs = (uint32_t)ops.size() * 2 * ctx->hc.k;
    if (type == DOP_ADD || type == DOP_SUB) hipLaunchKernelGGL(k_addsub_tab, dim3(limbs * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, dt, ctx->dc, ctx->chunks, type == DOP_SUB);
    else hipLaunchKernelGGL(k_add_plain_tab, dim3(limbs * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, dt, ctx->dc, ctx->chunks, type == DOP_SUBPLAIN);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    return 0;
}
static uint32_t chunk_for(cn_ctx *ctx, size_t per_ct, uint32_t count);
// all queued Multiply + Relinearize calls of one level: the batched BEHZ pipeline + ONE key switch, operands and results through tables
static int flush_mulrelin_group(cn_ctx *ctx, const std::vector<const DOp *> &all) {
    const size_t kn = (size_t)ctx->hc.k * ctx->hc.n;
    for (int sq = 1; sq >= 0; sq--) {                    // squarings (SquareActivation) take the fused kernel; general products the separate launches
        std::vector<const DOp *> ops;
        for (const DOp *op : all) if ((op->a == op->b) == (sq == 1)) ops.push_back(op);
        if (ops.empty()) continue;
        const size_t per = mul_scratch_per_ct(ctx, sq == 1) + al(3 * kn * 8) + 3 * 8 + 64;
        const uint32_t ch = chunk_for(ctx, per, (uint32_t)ops.size());
        for (uint32_t s0 = 0; s0 < ops.size(); s0 += ch) {
            const uint32_t c = std::min<uint32_t>(ch, (uint32_t)ops.size() - s0);
            CHECK(ensure_scratch(ctx, per * c + 3 * al((size_t)c * 8) + 8192));
            std::vector<const uint64_t *> ha(c), hb(c); std::vector<uint64_t *> ho(c);
            for (uint32_t i = 0; i < c; i++) { ha[i] = ops[s0 + i]->a; hb[i] = ops[s0 + i]->b; ho[i] = ops[s0 + i]->out; }
            const uint64_t **da, **db = nullptr; uint64_t **dout;
            CHECK(upload_tmp(ctx, ha.data(), c, &da));
            if (!sq) CHECK(upload_tmp(ctx, hb.data(), c, &db));
            CHECK(upload_tmp(ctx, ho.data(), c, &dout));
            uint64_t *t3 = salloc<uint64_t>(ctx, (size_t)c * 3 * kn);
            if (!t3) return fail(CN_ERR_HIP, "internal: scratch exhausted in deferred multiply");
            CHECK(do_multiply(ctx, nullptr, 1, nullptr, 1, t3, c, da, sq ? da : db));
            CHECK(do_keyswitch(ctx, t3 + 2 * kn, 3 * kn, t3, t3 + kn, 3 * kn, ctx->rlk, nullptr, c, 0, nullptr, 0, dout));
        }
    }
    return 0;
}

// ---- staged kinds (DOP_COPY .. DOP_SUMSLOTS): the per-ciphertext calls of an unchanged LoLa-style caller - one MultiplyPlain, SumAllSlots,
// RotateRows(AndAdd) per matrix row (EncryptedSealBfvMatrix.cs:79-120: `leVectors[row].DotProduct(v)` in a loop over the rows, LLInterleaveLayer:
// one PointwiseMultiply per column).  The rows are independent, so the queue puts row r's k-th call and row r''s k-th call on the same
// level; at flush all calls of one level, kind and parameter (rotation steps / slot count) are executed as ONE batched call of the same
// implementation the batched entry points use: their operand ciphertexts are gathered into a contiguous staging array (one table-driven
// copy launch), the batched implementation runs on it, the results are scattered to the callers' arrays (one more copy launch).  The
// copies move 2 x 640 KiB per ciphertext and call - microseconds against the key switches they let merge (13 rows of LoLa's dense
// layer: 13 x 10 single-ciphertext key switches become 10 key switches of 13 ciphertexts).  Same words: the batched implementations
// are bit-identical to their count-1 selves (tests/test_deferred.py, tests/test_lola.py).

__global__ void k_copy_tab(const Tab2 *__restrict__ tab, uint32_t pairs_per_item) {          // grid (chunks, items); 16 B per thread
    const Tab2 t = tab[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));          // (a plain vector type: assignable through a global-address-space pointer)
    if (i < pairs_per_item) reinterpret_cast<NTT_GLOBAL v2u64 *>(t.dst)[i] = reinterpret_cast<const NTT_GLOBAL v2u64 *>(t.src)[i];
}
static int ensure_stage(cn_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->stage_cap) return 0;
    if (ctx->capturing || ctx->graphs_alive) return fail(CN_ERR_ARG, "the staging arena would have to grow while a graph is recorded / alive");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->stage) HIPCHK(hipFree(ctx->stage));
    ctx->stage = nullptr; ctx->stage_cap = 0;
    const size_t want = bytes + (bytes >> 2) + (1 << 20);
    HIPCHK(hipMalloc((void **)&ctx->stage, want));
    ctx->stage_cap = want;
    return 0;
}
// host table -> device (a block the context keeps alive until the stream has drained, like upload_tmp), then one copy launch
static int copy_by_table(cn_ctx *ctx, const std::vector<Tab2> &tab, Tab2 *dtab, uint32_t words_per_item) {
    if (tab.empty()) return 0;
    const void *ptab;
    CHECK(place_table(ctx, tab.data(), tab.size() * sizeof(Tab2), dtab, &ptab));
    const uint32_t pairs = words_per_item / 2;
    hipLaunchKernelGGL(k_copy_tab, dim3((pairs + 255) / 256, (unsigned)tab.size()), dim3(256), 0, ctx->stream, (const Tab2 *)ptab, pairs);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    return 0;
}
static int flush_staged_group(cn_ctx *ctx, const std::vector<const DOp *> &all, int type) {
    if (type == DOP_COPY) {                                  // Ciphertext copies: the table copy is the operation
        CHECK(ensure_stage(ctx, al(all.size() * sizeof(Tab2))));
        std::vector<Tab2> tab(all.size());
        for (size_t i = 0; i < all.size(); i++) tab[i] = {(const NTT_GLOBAL uint64_t *)all[i]->a, (NTT_GLOBAL uint64_t *)all[i]->out};
        return copy_by_table(ctx, tab, (Tab2 *)ctx->stage, (uint32_t)ctx->ctw2);
    }
    if (type == DOP_ROT && all.size() * ctx->hc.k <= KS_WIDE_MAX_BLOCKS) {     // few rotations, any step counts: one launch chain per hop round
        std::vector<RotJob> jobs(all.size());
        for (size_t i = 0; i < all.size(); i++) jobs[i] = {all[i]->a, all[i]->out, (int)all[i]->arg, {}};
        return rotate_jobs(ctx, jobs);
    }
    std::map<int64_t, std::vector<const DOp *>> by_arg;      // one batched call per parameter value (rotation steps, slot count)
    for (const DOp *op : all) by_arg[op->arg].push_back(op);
    const uint32_t n = ctx->hc.n; const size_t ctw = ctx->ctw2;
    for (auto &kv : by_arg) {
        const std::vector<const DOp *> &ops = kv.second;
        const uint32_t cnt = (uint32_t)ops.size();
        const bool has_b = type == DOP_ROTADD || type == DOP_COLSADD, has_p = type == DOP_MULPLAIN, in_place = type == DOP_SUMSLOTS;
        const size_t tabs = al(3 * cnt * sizeof(Tab2)) + al(cnt * sizeof(Tab2)), ctb = al(cnt * ctw * 8);
        CHECK(ensure_stage(ctx, tabs + ctb * (1 + (has_b ? 1 : 0) + (in_place ? 0 : 1)) + (has_p ? al((size_t)cnt * n * 8) : 0)));
        char *base = ctx->stage;
        Tab2 *t_in = (Tab2 *)base, *t_pt = t_in + 2 * cnt, *t_out = (Tab2 *)(base + al(3 * cnt * sizeof(Tab2)));
        uint64_t *A = (uint64_t *)(base + tabs), *B = has_b ? A + ctb / 8 : nullptr;
        uint64_t *O = in_place ? A : A + (ctb / 8) * (has_b ? 2 : 1), *P = has_p ? O + ctb / 8 : nullptr;
        // an operand whose addresses are equally spaced already IS the array the batched implementation wants (always so for a single call -
        // 12 rotations by 12 different step counts are 12 groups of one): it is used in place; only scattered operands are gathered
        auto spaced = [&](auto get, size_t words) { for (uint32_t i = 1; i < cnt; i++) if (get(ops[i]) != get(ops[0]) + (size_t)i * words) return false; return true; };
        const bool da = spaced([](const DOp *o) { return o->a; }, ctw), db = has_b && spaced([](const DOp *o) { return o->b; }, ctw),
                   dp = has_p && spaced([](const DOp *o) { return o->b; }, n), dout = spaced([](const DOp *o) { return (const uint64_t *)o->out; }, ctw);
        std::vector<Tab2> gin, gpt, gout;
        for (uint32_t i = 0; i < cnt; i++) {
            if (!da) gin.push_back({(const NTT_GLOBAL uint64_t *)ops[i]->a, (NTT_GLOBAL uint64_t *)(A + (size_t)i * ctw)});
            if (has_b && !db) gin.push_back({(const NTT_GLOBAL uint64_t *)ops[i]->b, (NTT_GLOBAL uint64_t *)(B + (size_t)i * ctw)});
            if (has_p && !dp) gpt.push_back({(const NTT_GLOBAL uint64_t *)ops[i]->b, (NTT_GLOBAL uint64_t *)(P + (size_t)i * n)});
            if (!dout) gout.push_back({(const NTT_GLOBAL uint64_t *)(O + (size_t)i * ctw), (NTT_GLOBAL uint64_t *)ops[i]->out});
        }
        if (in_place && da != dout) return fail(CN_ERR_ARG, "internal: in-place staged call with different operand and result addresses");
        if (da) A = const_cast<uint64_t *>(ops[0]->a);
        if (db) B = const_cast<uint64_t *>(ops[0]->b);
        if (dp) P = const_cast<uint64_t *>(ops[0]->b);
        if (dout) O = ops[0]->out;
        if (!gin.empty()) CHECK(copy_by_table(ctx, gin, t_in, (uint32_t)ctw));
        if (!gpt.empty()) CHECK(copy_by_table(ctx, gpt, t_pt, n));
        Buffer fa, fb, fo, fp;
        auto fake = [&](Buffer &b, int kind, uint64_t *d, size_t item) { b.kind = kind; b.count = cnt; b.size = kind == 0 ? 2 : 1; b.d = d; b.item_words = item; };
        fake(fa, 0, A, ctw); fake(fb, 0, B, ctw); fake(fo, 0, O, ctw); fake(fp, 1, P, n);
        if (has_p) fp.pt_zero.assign(cnt, 0);                 // zero plaintexts were refused when the calls were queued
        int rc = 0;
        switch (type) {
        case DOP_MULPLAIN: rc = mul_plain_impl(ctx, &fa, 0, false, &fp, 0, 1, &fo, 0, cnt); break;
        case DOP_ROT: rc = rotate_rows_impl(ctx, &fa, 0, (int)kv.first, &fo, 0, cnt); break;
        case DOP_ROTADD: rc = rotate_rows_add_impl(ctx, &fa, 0, (int)kv.first, &fb, 0, &fo, 0, cnt); break;
        case DOP_COLS: rc = galois_impl(ctx, &fa, 0, 2ull * n - 1, &fo, 0, cnt); break;
        case DOP_COLSADD: rc = rotate_columns_add_impl(ctx, &fa, 0, &fb, 0, &fo, 0, cnt); break;
        case DOP_SUMSLOTS: rc = sum_slots_impl(ctx, &fa, 0, cnt, (uint32_t)kv.first); break;
        default: rc = fail(CN_ERR_ARG, "internal: staged kind %d", type);
        }
        CHECK(rc);
        if (!gout.empty()) CHECK(copy_by_table(ctx, gout, t_out, (uint32_t)ctw));
    }
    return 0;
}
// queue `count` per-ciphertext operations of a staged kind (arguments were checked by the caller)
static int defer_staged(cn_ctx *ctx, int type, Buffer *A, uint32_t ai, Buffer *B, uint32_t bi, const uint64_t *plain, uint32_t pstride_words, Buffer *O, uint32_t oi,
                        uint32_t count, int64_t arg) {
    for (uint32_t c = 0; c < count; c++) {
        const uint64_t *pa = A->d + (size_t)(ai + c) * A->item_words, *pb = B ? B->d + (size_t)(bi + c) * B->item_words : nullptr;
        DOp op{type, 0, O->d + (size_t)(oi + c) * O->item_words, pa, plain ? plain + (size_t)c * pstride_words : pb, 0, 0, nullptr};
        op.arg = arg;
        const uint64_t *ins[2] = {pa, pb};
        CHECK(defer_push(ctx, op, ins, 2));
    }
    switch (type) {
    case DOP_MULPLAIN: ctx->st.PlainMultiplication += 0; break;          // (counted by the batched implementation at flush time)
    default: break;
    }
    return 0;
}

// n single ciphertexts (or dense plaintexts) that live in n arrays into consecutive places of one array, ONE launch (see include/cnhip.h)
extern "C" int cn_copy_many(cn_ctx *ctx, const cn_handle *src, const uint32_t *sfirst, uint32_t n, cn_handle dst, uint32_t dfirst) { API_BODY
    LOCK_ONLY;
    if (!n) return 0;
    if (!src) return fail(CN_ERR_ARG, "null argument");
    Buffer *d = ctx->bufs.find(dst);
    if (!d || d->kind > 1) return fail(CN_ERR_ARG, "invalid handle");
    if (!range_ok(d, dfirst, n)) return fail(CN_ERR_ARG, "index out of range");
    std::vector<Buffer *> sb(n);
    for (uint32_t i = 0; i < n; i++) {
        Buffer *b = ctx->bufs.find(src[i]);
        const uint32_t f = sfirst ? sfirst[i] : 0;
        if (!b) return fail(CN_ERR_ARG, "invalid handle");
        if (b->kind != d->kind || b->item_words != d->item_words) return fail(CN_ERR_ARG, "copy between different buffer shapes");
        if (!range_ok(b, f, 1)) return fail(CN_ERR_ARG, "index out of range");
        if (b == d && f >= dfirst && f < dfirst + n && f != dfirst + i) return fail(CN_ERR_ARG, "copy_many: a source lies inside the destination range");
        sb[i] = b;
    }
    if (deferring(ctx) && d->kind == 0 && d->size == 2) {          // queued like n cn_copy calls
        for (uint32_t i = 0; i < n; i++) CHECK(defer_staged(ctx, DOP_COPY, sb[i], sfirst ? sfirst[i] : 0, nullptr, 0, nullptr, 0, d, dfirst + i, 1, 0));
        return 0;
    }
    CHECK(cn_defer_flush(ctx));
    CHECK(ensure_stage(ctx, al(n * sizeof(Tab2))));
    std::vector<Tab2> tab(n);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t f = sfirst ? sfirst[i] : 0;
        tab[i] = {(const NTT_GLOBAL uint64_t *)(sb[i]->d + (size_t)f * d->item_words), (NTT_GLOBAL uint64_t *)(d->d + (size_t)(dfirst + i) * d->item_words)};
        if (d->kind == 1) d->pt_zero[dfirst + i] = sb[i]->pt_zero[f];
    }
    return copy_by_table(ctx, tab, (Tab2 *)ctx->stage, (uint32_t)d->item_words);
API_END }

// all queued Encryptor.Encrypt calls of one level: one sampling / transform / tail launch chain over a table
static int flush_encrypt_group(cn_ctx *ctx, const std::vector<const DOp *> &ops) {
    std::vector<EncTab> tab(ops.size());
    for (size_t i = 0; i < ops.size(); i++) tab[i] = {(NTT_GLOBAL uint64_t *)ops[i]->out, (const NTT_GLOBAL uint64_t *)ops[i]->a, ops[i]->nonce, ops[i]->item};
    const size_t per = (size_t)ctx->hc.k * ctx->hc.n * 8 + 3 * (size_t)ctx->hc.n + sizeof(EncTab) + 64;
    const uint32_t ch = chunk_for(ctx, per, (uint32_t)ops.size());
    for (uint32_t s0 = 0; s0 < ops.size(); s0 += ch) {
        const uint32_t c = std::min<uint32_t>(ch, (uint32_t)ops.size() - s0);
        CHECK(encrypt_chain(ctx, c, nullptr, 0, nullptr, 0, tab.data() + s0));
    }
    return 0;
}
// the weighted sums of folded zero encryptions (cn_defer_flush), added onto the outputs of the scalar products `gemms` (launched just before): the samplers draw
// u, e1, e2 of every folded encryption exactly as flush_encrypt_group would have (its nonce, its item), k_encrypt_fold does the rest
static bool zero_fold_ok(cn_ctx *ctx) {
    return ctx->pk && ctx->enc_fused && !ctx->legacy_ntt && ctx->use_f64 && ctx->hc.q_f64 && ctx->hc.logn >= 10 && ctx->hc.logn <= 13;
}
static int flush_zero_folds(cn_ctx *ctx, DeferQueue *q, const std::vector<const DOp *> &gemms) {
    const uint32_t n = ctx->hc.n;
    std::vector<EncTab> tab; std::vector<FoldOut> fo; std::vector<FoldTerm> ft;
    const uint64_t t = ctx->hc.t.q, t_half = ctx->hc.t_half;
    for (const DOp *G : gemms) {
        fo.push_back({G->out, (uint32_t)ft.size(), G->fold_count});
        for (uint32_t f = 0; f < G->fold_count; f++) {
            const DeferQueue::Fold &fd = q->folds[(size_t)G->fold_first + f];
            const DOp &E = q->ops[fd.enc];
            tab.push_back({nullptr, nullptr, E.nonce, E.item});
            ft.push_back({fd.w >= t_half ? -(double)(t - fd.w) : (double)fd.w, (uint32_t)tab.size() - 1, 0});
        }
    }
    const uint32_t cnt = (uint32_t)tab.size();
    CHECK(ensure_scratch(ctx, al((size_t)cnt * n) + al((size_t)cnt * 2 * n) + al(cnt * sizeof(EncTab)) + al(fo.size() * sizeof(FoldOut)) + al(ft.size() * sizeof(FoldTerm)) + 1024));
    int8_t *us = salloc<int8_t>(ctx, (size_t)cnt * n), *es = salloc<int8_t>(ctx, (size_t)cnt * 2 * n);
    if (!us || !es) return fail(CN_ERR_HIP, "internal: scratch exhausted in the zero-encryption fold");
    EncTab *dtab; FoldOut *dfo; FoldTerm *dft;
    CHECK(upload_tmp(ctx, tab.data(), tab.size(), &dtab)); CHECK(upload_tmp(ctx, fo.data(), fo.size(), &dfo)); CHECK(upload_tmp(ctx, ft.data(), ft.size(), &dft));
    const RngKey key = rng_key_of(ctx);
    hipLaunchKernelGGL(k_sample_small, dim3((unsigned)(((uint64_t)cnt * (n / 16) + 255) / 256)), dim3(256), 0, ctx->stream, us, n, 0, 1u, cnt, key, 0ull, 0u, 0ull, (const EncTab *)dtab);
    hipLaunchKernelGGL(k_sample_small, dim3((unsigned)(((uint64_t)cnt * 2 * (n / 8) + 255) / 256)), dim3(256), 0, ctx->stream, es, n, 1, 2u, cnt, key, 0ull, 1u, 0ull, (const EncTab *)dtab);
    uint64_t qmax = 0; for (uint32_t j = 0; j < ctx->hc.k; j++) qmax = std::max(qmax, ctx->hc.q[j].q);
    if (!rr_ops[(qmax >> 44) ? POL_F64 : POL_F64L]->enc_fold(ctx, us, es, dfo, dft, (uint32_t)fo.size())) return fail(CN_ERR_ARG, "internal: zero-encryption fold without a kernel");
    HIPCHK(hipGetLastError()); launch_count(ctx, 3);
    ctx->st.ntt_forward_limbs += (uint64_t)fo.size() * ctx->hc.k; ctx->st.ntt_inverse_limbs += (uint64_t)fo.size() * 2 * ctx->hc.k;
    return 0;
}
static int defer_encrypt(cn_ctx *ctx, const uint64_t *ptd, uint32_t pt_stride_words, Buffer *O, uint32_t oi, uint32_t count, uint64_t seed) {
    for (uint32_t c = 0; c < count; c++) {
        DOp op{DOP_ENCRYPT, 0, O->d + (size_t)(oi + c) * O->item_words, ptd ? ptd + (size_t)c * pt_stride_words : nullptr, nullptr, 0, 0, nullptr};
        op.nonce = seed; op.item = ctx->rng_item++;
        CHECK(defer_push(ctx, op, nullptr, 0));
    }
    return 0;
}

static int cn_defer_flush(cn_ctx *ctx) {
    DeferQueue *q = ctx->dq;
    if (!q) return 0;
    int rc = 0;
    // CN_DEFER_TRACE=2: host time of every flush (the flush runs on the thread of the call that triggered it, under the context lock: every other caller of the
    // context waits for it, and so does the device if it has run dry)
    static const bool timing = getenv("CN_DEFER_TRACE") && atoi(getenv("CN_DEFER_TRACE")) >= 2;
    struct FlushTimer { bool on; size_t nops; cn_ctx *c; std::chrono::steady_clock::time_point t0; ~FlushTimer() {
        if (on && nops) fprintf(stderr, "defer %p flush of %zu calls: %.0f us of host time\n", (void *)c, nops,
                                1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); } }
        ft{timing, q->ops.size(), ctx, std::chrono::steady_clock::now()};
    if (!q->ops.empty()) {
        std::vector<DOp> &ops = q->ops;
        // ---- an AddPlain that only adds the bias to a DenseMatrixBySparseVectorMultiply result the caller has already released
        // (PoolLayer.cs:184-186: `using (conv = ConvolveOnce(..)) res[k] = conv.Add(bias)`) is folded into the GEMM's epilogue - the GEMM
        // then runs at the AddPlain's level and writes its output: safe when nothing else reads the intermediate and no later call
        // overwrites the GEMM's inputs
        std::vector<uint8_t> dead(ops.size(), 0);
        std::unordered_map<const uint64_t *, int> freed;
        for (auto &f : q->frees) freed[f.first] = 1;
        {
            for (size_t x = 0; x < ops.size(); x++) {
                DOp &X = ops[x];
                if (X.type != DOP_ADDPLAIN || X.a == X.out) continue;
                const DeferQueue::Haz *hi = q->haz.find(X.a);
                // the recorded writer must be the op that PRODUCED X's operand: a handle reused as the output of a later call has its last
                // writer BEHIND X (GEMM1 -> tmp, AddPlain(tmp) -> r1, GEMM2 -> tmp ...: folding GEMM2 into the first AddPlain would be wrong)
                if (!hi || hi->wop < 0 || hi->wop >= (int32_t)x || hi->readers != 1 || !freed.count(X.a)) continue;
                if (ops[hi->wop].level >= X.level) continue;
                DOp &Gm = ops[hi->wop];
                if (Gm.type != DOP_GEMM1 || Gm.bias || Gm.out != X.a || dead[hi->wop]) continue;
                bool ok = true;
                for (uint32_t kk = 0; kk < Gm.K && ok; kk++) {
                    const uint64_t *in = (const uint64_t *)q->addr[Gm.terms + kk];
                    if (!in) continue;
                    const DeferQueue::Haz *h2 = q->haz.find(in);
                    if (h2 && h2->wop > hi->wop) ok = false;
                }
                if (!ok) continue;
                Gm.out = X.out; Gm.bias = X.b; Gm.level = X.level;
                dead[x] = 1;
            }
        }
        // ---- fresh encryptions of ZERO that only feed one queued scalar product and have been released (PoolLayer.ElementAt / ReleaseTemp, PoolLayer.cs:67-90) are
        // not materialised: their weighted sum is folded onto the scalar product's output by linearity (k_encrypt_fold: same words, a fifth of the transforms, the
        // scalar product reads no extra ciphertexts and the outputs of a border patch share their gather list again).  Conditions, all on whole arrays: the
        // encryption is the last writer of its array, exactly one queued call reads it - a scalar product on a deeper level - and the caller has released it.
        if (ctx->fold_zero && zero_fold_ok(ctx)) {
            std::unordered_map<const uint64_t *, int32_t> cand;
            size_t zero_encs = 0;
            for (size_t x = 0; x < ops.size(); x++) {
                const DOp &E = ops[x];
                if (dead[x] || E.type != DOP_ENCRYPT || E.a) continue;
                zero_encs++;
                const DeferQueue::Haz *h = q->haz.find(E.out);
                if (h && h->wop == (int32_t)x && h->readers == 1 && freed.count(E.out)) cand[E.out] = (int32_t)x;
            }
            // all or nothing: a caller that parks its releases (cn_free_many of 32 at a time, the locked twin of rounds 3-5) leaves some of a layer's zero vectors alive at
            // the flush - folding the rest would run BOTH chains (samplers + k_encrypt_fused for the live ones, samplers + k_encrypt_fold for the others) and cut the scalar
            // products of a layer into more gather groups: 17.4 against 15.8 ms per batch (profiles/r06_bench_default_flags.json, `locked`)
            if (cand.size() != zero_encs) cand.clear();
            const uint64_t t_half = ctx->hc.t_half, max_terms = (1ull << 52) / std::max<uint64_t>(1, t_half * 20);
            if (!cand.empty()) for (size_t x = 0; x < ops.size(); x++) {
                DOp &G = ops[x];
                if (dead[x] || G.type != DOP_GEMM1 || G.level == 0) continue;
                uint32_t nf = 0, left = 0;
                for (uint32_t kk = 0; kk < G.K; kk++) {
                    const uint64_t a = q->addr[G.terms + kk];
                    if (!a) continue;
                    auto it = cand.find((const uint64_t *)a);
                    if (it != cand.end() && ops[it->second].level < G.level) nf++; else if (q->wt[G.terms + kk]) left++;
                }
                if (!nf || !left || nf > max_terms) continue;              // (a scalar product keeps at least one real term: its launch writes the output the fold adds onto)
                G.fold_first = (int32_t)q->folds.size();
                for (uint32_t kk = 0; kk < G.K; kk++) {
                    const uint64_t a = q->addr[G.terms + kk];
                    if (!a) continue;
                    auto it = cand.find((const uint64_t *)a);
                    if (it == cand.end() || ops[it->second].level >= G.level) continue;
                    if (q->wt[G.terms + kk]) { q->folds.push_back({it->second, q->wt[G.terms + kk]}); G.fold_count++; }      // (weight 0: the term contributes nothing, AtomicSealBfvVector.cs:468)
                    q->addr[G.terms + kk] = 0; q->wt[G.terms + kk] = 0;
                    dead[it->second] = 2;
                    cand.erase(it);
                }
                ctx->folded_zero += nf;
            }
        }
        // ---- scalar products of one term count on different levels of the same flush become ONE launch where nothing stands in the way (round 5).  The
        // literal PoolLayer pattern puts the outputs that read a fresh encryption of zero (a padded tap) one level behind the others: two GEMM launches per
        // convolution, the second one a fifth the size and barely half as efficient (k_scalar_gemm<1>: 0.30 ms for 125 outputs against 0.50 ms for 720).
        // A scalar product may wait for the deepest level that holds others of its kind if, behind its own level, nobody reads or writes its output and
        // nobody writes its inputs (checked against every queued call, conservatively: any later level counts).  CN_DEFER_MERGE_GEMM=0 switches it off (A/B).
        static const bool merge_gemm = !(getenv("CN_DEFER_MERGE_GEMM") && !atoi(getenv("CN_DEFER_MERGE_GEMM")));
        if (merge_gemm) {
            std::map<uint32_t, std::pair<int32_t, int32_t>> span;          // term count -> (shallowest, deepest) level of its live scalar products
            for (size_t x = 0; x < ops.size(); x++) if (!dead[x] && ops[x].type == DOP_GEMM1) {
                auto it = span.find(ops[x].K);
                if (it == span.end()) span[ops[x].K] = {ops[x].level, ops[x].level};
                else { it->second.first = std::min(it->second.first, ops[x].level); it->second.second = std::max(it->second.second, ops[x].level); }
            }
            bool any = false;
            for (auto &kv : span) any = any || kv.second.first != kv.second.second;
            if (any) {
                struct RW { int32_t r = -1, w = -1; };                      // deepest level at which a queued call reads / writes the array
                std::unordered_map<const uint64_t *, RW> touch;
                touch.reserve(ops.size() * 2);
                auto rd = [&](const uint64_t *p, int32_t lv) { if (p) { RW &t = touch[p]; t.r = std::max(t.r, lv); } };
                for (size_t x = 0; x < ops.size(); x++) {
                    if (dead[x]) continue;
                    const DOp &X = ops[x];
                    if (X.type == DOP_GEMM1) { for (uint32_t kk = 0; kk < X.K; kk++) rd((const uint64_t *)q->addr[X.terms + kk], X.level); }
                    else if (X.type == DOP_ADDPLAIN || X.type == DOP_SUBPLAIN || X.type == DOP_MULPLAIN) rd(X.a, X.level);      // (b is a plaintext: never the output of a queued call)
                    else if (X.type != DOP_ENCRYPT) { rd(X.a, X.level); rd(X.b, X.level); }
                    RW &t = touch[X.out]; t.w = std::max(t.w, X.level);
                }
                for (size_t x = 0; x < ops.size(); x++) {
                    DOp &X = ops[x];
                    if (dead[x] || X.type != DOP_GEMM1) continue;
                    const int32_t deep = span[X.K].second;
                    if (X.level >= deep) continue;
                    const RW &to = touch[X.out];
                    bool ok = to.r <= X.level && to.w <= X.level;
                    for (uint32_t kk = 0; kk < X.K && ok; kk++) {
                        const uint64_t *in = (const uint64_t *)q->addr[X.terms + kk];
                        if (in) { auto it = touch.find(in); ok = it == touch.end() || it->second.w <= X.level; }
                    }
                    if (ok) X.level = deep;
                }
            }
        }
        // ---- launches: level by level, one batched launch per kind (and per term count for the GEMMs)
        const int32_t levels = q->maxlevel + 1;
        static const bool trace = getenv("CN_DEFER_TRACE") && atoi(getenv("CN_DEFER_TRACE"));       // one line per (flush, level): calls per kind, launches
        for (int32_t lv = 0; lv < levels && !rc; lv++) {
            std::vector<const DOp *> by_type[DOP_TYPES];
            for (size_t x = 0; x < ops.size(); x++) if (!dead[x] && ops[x].level == lv) by_type[ops[x].type].push_back(&ops[x]);
            const uint64_t l0 = ctx->st.kernel_launches;
            struct Tr { cn_ctx *c; int32_t lv; uint64_t l0; std::vector<const DOp *> *bt; bool on; ~Tr() {
                if (!on) return;
                char line[512]; int o = snprintf(line, sizeof line, "defer %p level %d:", (void *)c, lv);
                for (int t = 0; t < DOP_TYPES; t++) if (!bt[t].empty()) {
                    std::map<int64_t, int> args; for (const DOp *op : bt[t]) args[op->arg]++;
                    o += snprintf(line + o, sizeof line - o, " kind%d x%zu (%zu args)", t, bt[t].size(), args.size());
                }
                fprintf(stderr, "%s -> %llu launches\n", line, (unsigned long long)(c->st.kernel_launches - l0)); } } tr{ctx, lv, l0, by_type, trace};
            if (!by_type[DOP_GEMM1].empty()) {
                std::map<uint32_t, std::vector<const DOp *>> byK;
                for (const DOp *op : by_type[DOP_GEMM1]) byK[op->K].push_back(op);
                for (auto &kv : byK) if (!rc) rc = flush_gemm_group(ctx, q, kv.second, kv.first);
                std::vector<const DOp *> folded;
                for (const DOp *op : by_type[DOP_GEMM1]) if (op->fold_count) folded.push_back(op);
                if (!rc && !folded.empty()) rc = flush_zero_folds(ctx, q, folded);
            }
            for (int t : {DOP_ADD, DOP_SUB, DOP_ADDPLAIN, DOP_SUBPLAIN}) if (!rc && !by_type[t].empty()) rc = flush_elementwise_group(ctx, by_type[t], t);
            if (!rc && !by_type[DOP_ENCRYPT].empty()) rc = flush_encrypt_group(ctx, by_type[DOP_ENCRYPT]);
            for (int t = DOP_COPY; t <= DOP_SUMSLOTS; t++) if (!rc && !by_type[t].empty()) rc = flush_staged_group(ctx, by_type[t], t);
            if (!rc && !by_type[DOP_MULRELIN].empty()) rc = flush_mulrelin_group(ctx, by_type[DOP_MULRELIN]);
        }
    }
    q->ops.clear(); q->addr.clear(); q->wt.clear(); q->haz.clear(); q->maxlevel = -1; q->folds.clear();
    for (auto &f : q->frees) { int r2 = dev_release(ctx, f.first, f.second); if (!rc) rc = r2; }
    q->frees.clear();
    return rc;
}
// true when the call is to be queued rather than launched (the caller holds the lock)
static bool deferring(cn_ctx *ctx) { return ctx->defer && !ctx->capturing; }

/* DenseMatrixBySparseVectorMultiply for ONE output block whose K input ciphertexts are separate objects (see include/cnhip.h) */
static int scalar_dot_body(cn_ctx *ctx, const cn_handle *in, const uint32_t *in_idx, const uint64_t *w, uint32_t K, cn_handle out, uint32_t oi);
extern "C" int cn_scalar_dot(cn_ctx *ctx, const cn_handle *in, const uint32_t *in_idx, const uint64_t *w, uint32_t K, cn_handle out, uint32_t oi) {
    if (submit_async(ctx) && K && in && w) {            // the call's lists travel in one block: K handles, K weights, K indices (if any)
        char *blk = (char *)malloc((size_t)K * (in_idx ? 20 : 16));
        if (!blk) return fail(CN_ERR_ARG, "out of host memory");
        memcpy(blk, in, (size_t)K * 8); memcpy(blk + (size_t)K * 8, w, (size_t)K * 8);
        if (in_idx) memcpy(blk + (size_t)K * 16, in_idx, (size_t)K * 4);
        return ring_push(ctx, SUB_SCALAR_DOT, 1, 0, in_idx ? 1u : 0u, 0, 0, out, oi, K, (uint64_t)(uintptr_t)blk);
    }
    API_BODY LOCK_ONLY; return scalar_dot_body(ctx, in, in_idx, w, K, out, oi); API_END
}
static int scalar_dot_body(cn_ctx *ctx, const cn_handle *in, const uint32_t *in_idx, const uint64_t *w, uint32_t K, cn_handle out, uint32_t oi) {
    GETCT(O, out, 2);
    if (!K || !in || !w) return fail(CN_ERR_ARG, "empty scalar product");
    if (oi >= O->count) return fail(CN_ERR_ARG, "index out of range");
    DeferQueue *q = ctx->dq;
    const size_t t0 = q->addr.size();
    const uint64_t *ins_small[64];                           // (no heap allocation per call for the usual window sizes: 25 taps)
    std::vector<const uint64_t *> ins_big;
    const uint64_t **ins_p = ins_small;
    if (K > 64) { ins_big.assign(K, nullptr); ins_p = ins_big.data(); } else for (uint32_t kk = 0; kk < K; kk++) ins_small[kk] = nullptr;
    struct InsView { const uint64_t **p; const uint64_t *&operator[](uint32_t i) { return p[i]; } const uint64_t **data() { return p; } } ins{ins_p};
    bool any = false;
    uint64_t *o = O->d + (size_t)oi * O->item_words;
    uint64_t nnz = 0;
    for (uint32_t kk = 0; kk < K; kk++) {
        const uint64_t wk = w[kk];
        if (wk >= ctx->hc.t.q) { q->addr.resize(t0); q->wt.resize(t0); return fail(CN_ERR_ARG, "weight >= plain modulus"); }
        uint64_t a = 0;
        if (in[kk]) {                                        // handle 0: padded tap (PoolLayer.cs:68-80), skipped
            Buffer *I = getbuf(ctx, in[kk], 0);
            const uint32_t ii = in_idx ? in_idx[kk] : 0;
            if (!I || I->size != 2 || ii >= I->count) { q->addr.resize(t0); q->wt.resize(t0); return fail(CN_ERR_ARG, "invalid input ciphertext %u", kk); }
            const uint64_t *p = I->d + (size_t)ii * I->item_words;
            if (p == o) { q->addr.resize(t0); q->wt.resize(t0); return fail(CN_ERR_ARG, "scalar product cannot run in place"); }
            a = (uint64_t)p; ins[kk] = p;                    // (a zero weight keeps its address: outputs that share a patch still share a gather list)
            if (wk) { any = true; nnz++; }                   // zero weights contribute nothing (AtomicSealBfvVector.cs:468 skips them)
        }
        q->addr.push_back(a); q->wt.push_back(a ? wk : 0);
    }
    if (!any) { q->addr.resize(t0); q->wt.resize(t0); return fail(CN_ERR_ARG, "output has no non-zero term (AddMany of nothing)"); }
    DOp op{DOP_GEMM1, 0, o, nullptr, nullptr, K, t0, nullptr};
    CHECK(defer_push(ctx, op, ins.data(), K));
    ctx->st.PlainMultiplication += nnz; ctx->st.Addition += nnz - 1;
    if (!deferring(ctx)) return cn_defer_flush(ctx);
    return 0;
}

// ---- the deferrable forms of the per-ciphertext entry points (arguments are checked now, the work is queued)
static int defer_addsub(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count, int op) {
    GETCT(A, a, 0); GETCT(O, out, A->size);
    Buffer *B = getbuf(ctx, b, 0);
    if (!B || B->size != A->size) return fail(CN_ERR_ARG, "operand sizes do not match");
    if (!range_ok(A, ai, count) || !range_ok(O, oi, count) || !range_ok(B, bi, count)) return fail(CN_ERR_ARG, "index out of range");
    if (A->size != 2) return 1;
    for (uint32_t c = 0; c < count; c++) {
        const uint64_t *pa = A->d + (size_t)(ai + c) * A->item_words, *pb = B->d + (size_t)(bi + c) * B->item_words;
        const uint64_t *ins[2] = {pa, pb};
        CHECK(defer_push(ctx, DOp{op ? DOP_SUB : DOP_ADD, 0, O->d + (size_t)(oi + c) * O->item_words, pa, pb, 0, 0, nullptr}, ins, 2));
    }
    if (op) ctx->st.Subtraction += count; else ctx->st.Addition += count;
    return 0;
}
static int defer_add_plain(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, int subtract, cn_handle out, uint32_t oi, uint32_t count) {
    GETCT(A, a, 0); GETCT(O, out, A->size); GETPT(P, pt);
    if (!range_ok(A, ai, count) || !range_ok(O, oi, count) || !range_ok(P, pi, count)) return fail(CN_ERR_ARG, "index out of range");
    if (A->size != 2) return 1;
    for (uint32_t c = 0; c < count; c++) {
        const uint64_t *pa = A->d + (size_t)(ai + c) * A->item_words;
        const uint64_t *ins[1] = {pa};
        CHECK(defer_push(ctx, DOp{subtract ? DOP_SUBPLAIN : DOP_ADDPLAIN, 0, O->d + (size_t)(oi + c) * O->item_words, pa, P->d + (size_t)(pi + c) * ctx->hc.n, 0, 0, nullptr}, ins, 1));
    }
    if (subtract) ctx->st.PlainSubtraction += count; else ctx->st.PlainAddition += count;
    return 0;
}
static int defer_mul_relin(cn_ctx *ctx, cn_handle a, uint32_t ai, uint32_t astride, cn_handle b, uint32_t bi, uint32_t bstride, cn_handle out, uint32_t oi, uint32_t count) {
    GETCT(A, a, 2); GETCT(B, b, 2); GETCT(O, out, 2);
    if (!range_ok(A, ai, astride ? count : 1, astride ? astride : 1) || !range_ok(B, bi, bstride ? count : 1, bstride ? bstride : 1) || !range_ok(O, oi, count))
        return fail(CN_ERR_ARG, "index out of range");
    if (!ctx->rlk.d) return fail(CN_ERR_NOKEY, "relinearization keys not set");
    for (uint32_t c = 0; c < count; c++) {
        const uint64_t *pa = A->d + ((size_t)ai + (size_t)c * astride) * A->item_words, *pb = B->d + ((size_t)bi + (size_t)c * bstride) * B->item_words;
        const uint64_t *ins[2] = {pa, pb};
        CHECK(defer_push(ctx, DOp{DOP_MULRELIN, 0, O->d + (size_t)(oi + c) * O->item_words, pa, pb, 0, 0, nullptr}, ins, 2));
    }
    ctx->st.Relinarization += count;          // (Multiplication is counted by the batched multiply at flush time)
    return 0;
}

// ---------------------------------------------------------------- lock-free submission ("defer" = 2, cn_submit.h): the consumer side
// ready_refill (lock held): single-ciphertext arrays for the lock-free cn_ct_alloc / cn_encrypt_zero_new.  The target starts small and doubles whenever
// a caller found the ring empty since the last refill (a flush that holds the lock for a millisecond is outrun by ~1 000 allocations).
static void ready_refill(cn_ctx *ctx) {
    ReadyRing &r = *ctx->ready;
    if (ctx->capturing) return;
    if (r.misses.exchange(0, std::memory_order_relaxed)) r.target = std::min<uint32_t>(r.target * 2, (uint32_t)ReadyRing::CAP / 2);
    while (r.size() < r.target) {
        cn_handle h = 0;
        if (alloc_buf(ctx, 0, 1, 2, &h)) return;                 // out of memory: the callers fall back to the locked path and see the error there
        if (!r.push(h)) { Buffer *b = ctx->bufs.find(h); (void)dev_release(ctx, b->d, b->item_words * 8); ctx->bufs.erase(h); return; }
    }
}
static int ring_exec(cn_ctx *ctx, const SubRec &r) {
    switch (r.type) {
    case SUB_FREE: return free_body(ctx, r.a);
    case SUB_FREE_MANY: { cn_handle *blk = (cn_handle *)(uintptr_t)r.arg; const int rc = free_many_body(ctx, blk, r.x); free(blk); return rc; }
    case SUB_SCALAR_DOT: {
        char *blk = (char *)(uintptr_t)r.arg; const uint32_t K = r.x;
        const int rc = scalar_dot_body(ctx, (const cn_handle *)blk, r.ai ? (const uint32_t *)(blk + (size_t)K * 16) : nullptr, (const uint64_t *)(blk + (size_t)K * 8), K, r.out, r.oi);
        free(blk);
        return rc;
    }
    case SUB_ADD: return addsub_body(ctx, r.a, r.ai, r.b, r.bi, r.out, r.oi, r.count, 0);
    case SUB_SUB: return addsub_body(ctx, r.a, r.ai, r.b, r.bi, r.out, r.oi, r.count, 1);
    case SUB_ADD_PLAIN: return add_plain_body(ctx, r.a, r.ai, r.b, r.bi, (int)r.x, r.out, r.oi, r.count);
    case SUB_MUL_RELIN: return mul_relin_body(ctx, r.a, r.ai, r.x, r.b, r.bi, (uint32_t)r.arg, r.out, r.oi, r.count);
    case SUB_ENCRYPT: return encrypt_body(ctx, r.b, r.bi, r.x, r.out, r.oi, r.count, r.arg);
    case SUB_ENCRYPT_ZERO: {
        Buffer *O = ctx->bufs.find(r.out);
        if (!O || O->kind != 0 || O->size != 2) return fail(CN_ERR_ARG, "invalid ciphertext handle");
        if (!ctx->pk) return fail(CN_ERR_NOKEY, "public key not set");
        if (ctx->hc.logn < 10 || ctx->hc.logn > 14) return fail(CN_ERR_ARG, "device encryption needs 1024 <= N <= 16384");
        return defer_encrypt(ctx, nullptr, 0, O, 0, 1, r.arg);
    }
    default: return fail(CN_ERR_ARG, "internal: submission record of type %u", r.type);
    }
}
// lock held.  Executes published records in claim order; upto = ~0: as far as they are published (an opportunistic drain stops at a slot that is claimed
// but not written yet), else every record claimed before position `upto` (waiting for a writer that was descheduled between its claim and its publication).
static void ring_drain(cn_ctx *ctx, uint64_t upto) {
    SubmitRing &q = *ctx->ring;
    (void)hipSetDevice(ctx->device);
    uint64_t done = 0;
    for (;;) {
        SubRec *r = q.peek();
        if (!r) {
            if (upto == ~0ull || q.head.load(std::memory_order_relaxed) >= upto) break;
            for (int spins = 0; !(r = q.peek()); spins++) { if (spins < 256) __builtin_ia32_pause(); else sched_yield(); }
        }
        const int rc = ring_exec(ctx, *r);
        if (rc && !ctx->async_rc) { ctx->async_rc = rc; ctx->async_msg = cn_last_error(); }
        q.pop();
        done++;
    }
    if (done && ctx->defer.load(std::memory_order_relaxed) == 2) ready_refill(ctx);
}
static int ring_sync(cn_ctx *ctx, bool report) {
    SubmitRing &q = *ctx->ring;
    const uint64_t t = q.tail.load(std::memory_order_acquire);
    if (q.head.load(std::memory_order_relaxed) != t) ring_drain(ctx, t);
    if (report && ctx->async_rc) {
        const int rc = ctx->async_rc; ctx->async_rc = 0;
        return fail(rc, "a call submitted without the lock (defer = 2) failed when it was executed: %s", ctx->async_msg.c_str());
    }
    return 0;
}
// producer: claim, write, publish; then drain if nobody else is (nobody waits for the lock here)
static int ring_push(cn_ctx *ctx, uint32_t type, uint32_t count, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t x, uint64_t arg) {
    SubmitRing &q = *ctx->ring;
    const uint64_t pos = q.claim();
    for (int spins = 0; !q.writable(pos); spins++) {           // a full lap ahead of the consumer: help
        if (ctx->mu.try_lock()) { ring_drain(ctx, ~0ull); ctx->mu.unlock_now(); }
        else if (spins < 64) __builtin_ia32_pause(); else sched_yield();
    }
    SubRec &r = q.slot(pos);
    r.type = type; r.count = count; r.a = a; r.b = b; r.out = out; r.ai = ai; r.bi = bi; r.oi = oi; r.x = x; r.arg = arg;
    q.publish(pos);
    while (q.peek_published() && ctx->mu.try_lock()) { ring_drain(ctx, ~0ull); ctx->mu.unlock_now(); }
    return 0;
}

// ---------------------------------------------------------------- multi-GPU: evaluation keys of ctxs[0] to every other context
// The path shards by independent batches / plaintext primes: the only exchange is this one-time key broadcast.  A host that
// runs one process per GPU (bench.py) broadcasts with torch.distributed and adopts the buffers (cn_set_relin_key, is_device_ptr = 1); a
// single-process multi-threaded host (the C# one) calls this: contexts on OTHER devices receive the keys with ONE RCCL broadcast per key
// over xGMI (librccl is loaded on demand; without it: peer copies), contexts on the root's device with device-to-device copies.
#include <dlfcn.h>
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast"); GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast) { dlclose(lib); lib = nullptr; return false; }
        return true;
    }
};
const int kNcclUint64 = 5;          // ncclDataType_t
}
#define NCCLCHK(x) do { int r_ = (x); if (r_) { rc = fail(CN_ERR_HIP, "%s failed: %s", #x, R.GetErrorString ? R.GetErrorString(r_) : "rccl error"); goto done; } } while (0)
extern "C" int cn_ctx_broadcast_keys(cn_ctx **ctxs, int n) {
    if (!ctxs || n < 1 || !ctxs[0]) return fail(CN_ERR_ARG, "null argument");
    cn_ctx *root = ctxs[0];
    for (int i = 1; i < n; i++) {
        cn_ctx *c = ctxs[i];
        if (!c || c == root) return fail(CN_ERR_ARG, "context %d is null or the root itself", i);
        bool same = c->hc.n == root->hc.n && c->hc.k == root->hc.k && c->hc.t.q == root->hc.t.q && c->hc.dbc == root->hc.dbc && c->hc.gdbc == root->hc.gdbc;
        for (uint32_t j = 0; same && j < root->hc.k; j++) same = c->hc.q[j].q == root->hc.q[j].q;
        if (!same) return fail(CN_ERR_ARG, "context %d has other encryption parameters than the root", i);
    }
    // everything in flight on the contexts is finished first; then the ROOT's lock is held for the whole broadcast (its key table is
    // read and its key buffers are the sources: a concurrent cn_set_galois_key on the root must not free or re-map them meanwhile).
    // Callers broadcast at start-up, from one thread, with one root.
    for (int i = 0; i < n; i++) CHECK(cn_sync(ctxs[i]));
    CnGuard root_lock(root->mu);
    struct Item { uint64_t elt; bool galois; KsKey src; size_t words; };
    std::vector<Item> items;
    if (root->rlk.d) items.push_back({0, false, root->rlk, cn_key_words(root, 0)});
    for (auto &kv : root->gk) if (kv.second.d) items.push_back({kv.first, true, kv.second, cn_key_words(root, 1)});
    if (items.empty()) return fail(CN_ERR_NOKEY, "the root context has no evaluation keys");
    // destination buffers
    std::vector<std::vector<uint64_t *>> dst(n, std::vector<uint64_t *>(items.size(), nullptr));
    int rc = 0;
    Rccl R;
    std::vector<int> devs;                   // distinct devices, the root's first; leader[d] = first context on devs[d]
    std::vector<int> leader;
    std::vector<void *> comms;
    auto dev_index = [&](int device) { for (size_t d = 0; d < devs.size(); d++) if (devs[d] == device) return (int)d; return -1; };
    for (int i = 0; i < n; i++) if (dev_index(ctxs[i]->device) < 0) { devs.push_back(ctxs[i]->device); leader.push_back(i); }
    for (int i = 1; i < n && !rc; i++) {
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) { rc = fail(CN_ERR_HIP, "hipSetDevice failed"); break; }
        for (size_t x = 0; x < items.size(); x++)
            if (hipMalloc((void **)&dst[i][x], items[x].words * 8) != hipSuccess) { rc = fail(CN_ERR_HIP, "out of device memory for the broadcast keys"); break; }
    }
    const bool force = getenv("CN_BCAST_FORCE_RCCL") && atoi(getenv("CN_BCAST_FORCE_RCCL"));
    const bool use_rccl = !rc && (devs.size() > 1 || force) && R.load();
    if (!rc && use_rccl) {
        comms.assign(devs.size(), nullptr);
        NCCLCHK(R.CommInitAll(comms.data(), (int)devs.size(), devs.data()));
        for (size_t x = 0; x < items.size(); x++) {
            NCCLCHK(R.GroupStart());
            for (size_t d = 0; d < devs.size(); d++) {
                cn_ctx *c = ctxs[leader[d]];
                void *buf = d == 0 ? (void *)items[x].src.d : (void *)dst[leader[d]][x];
                if (hipSetDevice(c->device) != hipSuccess) { rc = fail(CN_ERR_HIP, "hipSetDevice failed"); goto done; }
                NCCLCHK(R.Broadcast(buf, buf, items[x].words, kNcclUint64, 0, comms[d], c->stream));
            }
            NCCLCHK(R.GroupEnd());
        }
    }
    for (int i = 1; i < n && !rc; i++) {     // contexts that did not receive through RCCL: copies from their device's leader (or from the root)
        const int d = dev_index(ctxs[i]->device);
        const bool got = use_rccl && leader[d] == i;
        if (got) continue;
        const int from = (use_rccl || d == 0) ? leader[d] : 0;
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) { rc = fail(CN_ERR_HIP, "hipSetDevice failed"); break; }
        if (from != 0 && hipStreamSynchronize(ctxs[from]->stream) != hipSuccess) { rc = fail(CN_ERR_HIP, "synchronisation failed"); break; }
        for (size_t x = 0; x < items.size() && !rc; x++) {
            const void *src = from == 0 ? (const void *)items[x].src.d : (const void *)dst[from][x];
            hipError_t e = ctxs[from]->device == ctxs[i]->device ? hipMemcpyAsync(dst[i][x], src, items[x].words * 8, hipMemcpyDeviceToDevice, ctxs[i]->stream)
                                                                 : hipMemcpyPeerAsync(dst[i][x], ctxs[i]->device, src, ctxs[from]->device, items[x].words * 8, ctxs[i]->stream);
            if (e != hipSuccess) rc = fail(CN_ERR_HIP, "key copy failed: %s", hipGetErrorString(e));
        }
    }
done:
    for (int i = 0; i < n; i++) { (void)hipSetDevice(ctxs[i]->device); (void)hipStreamSynchronize(ctxs[i]->stream); }
    for (void *cm : comms) if (cm) (void)R.CommDestroy(cm);
    for (int i = 1; i < n; i++) {
        CnGuard lk(ctxs[i]->mu);
        // the keys are of ONE decomposition convention (cn_set_option("ks_xi"), settled per context by the client's start-up self-test): a replica that adopts the
        // root's keys adopts its convention with them - with the other one every Relinearize / Rotate would return rc 0 and garbage
        if (!rc && ctxs[i]->hc.ks_xi != root->hc.ks_xi) {
            if (ctxs[i]->capturing || ctxs[i]->graphs_alive) rc = fail(CN_ERR_ARG, "context %d holds recorded graphs of the other key-switch convention", i);
            else {
                (void)hipSetDevice(ctxs[i]->device);
                ctxs[i]->hc.ks_xi = root->hc.ks_xi;
                if (hipMemcpy(ctxs[i]->dc, &ctxs[i]->hc, sizeof(DevConsts), hipMemcpyHostToDevice) != hipSuccess) rc = fail(CN_ERR_HIP, "constant upload failed on context %d", i);
                // the keys this replica already holds were made for the OTHER convention: those the broadcast does not overwrite are dropped (a rotation by one of
                // their elements then fails with CN_ERR_NOKEY instead of returning rc 0 and garbage - ADVICE r05)
                if (!rc) {
                    auto carried = [&](bool galois, uint64_t elt) { for (const Item &it : items) if (it.galois == galois && (!galois || it.elt == elt)) return true; return false; };
                    if (ctxs[i]->rlk.d && !carried(false, 0)) { if (ctxs[i]->rlk.owned) (void)hipFree(ctxs[i]->rlk.d); ctxs[i]->rlk = KsKey{nullptr, false, false}; }
                    for (auto it = ctxs[i]->gk.begin(); it != ctxs[i]->gk.end();) {
                        if (!carried(true, it->first)) { if (it->second.owned && it->second.d) (void)hipFree(it->second.d); it = ctxs[i]->gk.erase(it); } else ++it;
                    }
                }
            }
        }
        for (size_t x = 0; x < items.size(); x++) {
            if (!dst[i][x]) continue;
            if (rc) { (void)hipSetDevice(ctxs[i]->device); (void)hipFree(dst[i][x]); continue; }
            KsKey &slot = items[x].galois ? ctxs[i]->gk[items[x].elt] : ctxs[i]->rlk;
            if (slot.owned && slot.d) { (void)hipSetDevice(ctxs[i]->device); (void)hipFree(slot.d); }
            slot = KsKey{dst[i][x], true, items[x].src.f64};           // the words arrive in the form the root keeps them (FP64 image or u64) ...
            const bool want = keys_as_f64(ctxs[i]);                    // ... and are converted when this context keeps the other form (f64 = 0, legacy_ntt)
            if (want != slot.f64) {
                (void)hipSetDevice(ctxs[i]->device);
                const size_t words = items[x].words;
                if (want) hipLaunchKernelGGL(k_u64_to_f64, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctxs[i]->stream, slot.d, words);
                else hipLaunchKernelGGL(k_f64_to_u64, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctxs[i]->stream, slot.d, words);
                if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctxs[i]->stream) != hipSuccess) rc = fail(CN_ERR_HIP, "key conversion failed on context %d", i);
                slot.f64 = want;
            }
        }
    }
    (void)hipSetDevice(root->device);
    return rc;
}

// ---------------------------------------------------------------- raw transforms / timing / stats
static int raw_ntt(cn_ctx *ctx, void *p, uint32_t limbs, int base, int inverse) {
    if (base != 0 && base != 1) return fail(CN_ERR_ARG, "base must be 0 (q) or 1 (Bsk)");
    return cn_run_ntt(ctx, (uint64_t *)p, limbs, base ? ctx->hc.k : 0, base ? ctx->hc.kb : ctx->hc.k, inverse);
}
extern "C" int cn_ntt_forward(cn_ctx *ctx, void *p, uint32_t limbs, int base) { API_BODY LOCK; return raw_ntt(ctx, p, limbs, base, 0); API_END }
extern "C" int cn_ntt_inverse(cn_ctx *ctx, void *p, uint32_t limbs, int base) { API_BODY LOCK; return raw_ntt(ctx, p, limbs, base, 1); API_END }
extern "C" int cn_ct_ntt(cn_ctx *ctx, cn_handle h, uint32_t first, uint32_t count, int inverse) { API_BODY
    LOCK; GETCT(B, h, 0);
    if (!range_ok(B, first, count)) return fail(CN_ERR_ARG, "index out of range");
    return raw_ntt(ctx, B->d + first * B->item_words, count * B->size * ctx->hc.k, 0, inverse);
API_END }
extern "C" int cn_ntt_time(cn_ctx *ctx, void *p, uint32_t limbs, int base, int inverse, int iters, float *ms) { API_BODY
    LOCK; NOT_CAPTURING("cn_ntt_time");
    if (iters < 1 || !ms) return fail(CN_ERR_ARG, "bad arguments");
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < iters; i++) CHECK(raw_ntt(ctx, p, limbs, base, inverse));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipEventSynchronize(ctx->ev1));
    float t = 0; HIPCHK(hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
    *ms = t / iters;
    return 0;
API_END }
// ns per wave-instruction per SIMD RIGHT NOW (same occupancy as the fused key switch: 512-thread workgroups, one per CU, two waves per
// SIMD), over `launches` launches.  kind 0: FP64 (8 independent chains of the 6-instruction modular multiply per thread, `iters` x 48
// instructions); kind 1: full-rate 32-bit VALU (4 chains x 6 instructions, `iters` x 24)
extern "C" int cn_valu_issue_time(cn_ctx *ctx, int kind, int iters, int launches, float *ns_per_instr) { API_BODY
    LOCK; NOT_CAPTURING("cn_valu_issue_time");
    if (iters < 1 || launches < 1 || !ns_per_instr || kind < 0 || kind > 1) return fail(CN_ERR_ARG, "bad arguments");
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    const size_t lds = 96 * 1024;                                   // > half of the 160 KiB: one workgroup per CU
    HIPCHK(hipFuncSetAttribute((const void *)k_fp64_probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(hipFuncSetAttribute((const void *)k_valu_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(ensure_scratch(ctx, al((size_t)cus * 512 * 8)));
    double *out = salloc<double>(ctx, (size_t)cus * 512);
    const double q = 8796092792833.0, w = 1234567891011.0;
    auto launch = [&]() {
        if (kind == 0) hipLaunchKernelGGL(k_fp64_probe<8>, dim3(cus), dim3(512), lds, ctx->stream, out, w, q, 1.0 / q, iters);
        else hipLaunchKernelGGL(k_valu_probe, dim3(cus), dim3(512), lds, ctx->stream, out, 0x9e3779b9u, 0x7f4a7c15u, iters);
    };
    launch();
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < launches; i++) launch();
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(hipEventSynchronize(ctx->ev1));
    HIPCHK(hipGetLastError());
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *ns_per_instr = ms * 1e6f / ((float)launches * 2.0f * (float)iters * (kind == 0 ? 48.0f : 24.0f));       // 2 waves per SIMD
    return 0;
API_END }
extern "C" int cn_event_time_begin(cn_ctx *ctx) { API_BODY LOCK; HIPCHK(hipEventRecord(ctx->ev0, ctx->stream)); return 0; API_END }
extern "C" int cn_event_time_end(cn_ctx *ctx, float *ms) { API_BODY
    LOCK; NOT_CAPTURING("cn_event_time_end");
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream)); HIPCHK(hipEventSynchronize(ctx->ev1));
    HIPCHK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return 0;
API_END }
extern "C" int cn_stats_get(cn_ctx *ctx, cn_stats *out, int reset) { API_BODY
    LOCK;                                        // queued calls are launched first: Multiplication is counted by the batched multiply at flush time
    if (out) *out = ctx->st;
    if (reset) ctx->st = cn_stats{};
    return 0;
API_END }

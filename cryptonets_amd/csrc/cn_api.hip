// libcnhip.so host runtime (1/5): contexts, streams, options, device buffers, graphs, uploads / downloads, encoder, raw transforms, timing, statistics - behind the
// C ABI of include/cnhip.h.  The compute path is HIP-only: there is no CPU fallback; every entry point fails with CN_ERR_NODEV / CN_ERR_HIP when no gfx950 device is usable.
#include "cn_api_shared.h"

// ---------------------------------------------------------------- helpers
// deferred submission of per-ciphertext calls (second half of this file)
std::vector<Slab> &slabs_of(cn_ctx *ctx) { return *reinterpret_cast<std::vector<Slab> *>(ctx->slabs); }
bool in_slab(cn_ctx *ctx, const void *p) {
    for (const Slab &s : slabs_of(ctx)) if ((const char *)p >= s.base && (const char *)p < s.base + s.bytes) return true;
    return false;
}
int use(cn_ctx *c) { HIPCHK(hipSetDevice(c->device)); return 0; }

int ensure_scratch(cn_ctx *c, size_t bytes) {
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
size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
// Small host tables (gather indices, weight tiles) travel with an asynchronous copy on the context stream.  The caller's buffer is
// usually a local std::vector, so the bytes are first moved into a staging block the context keeps alive until the stream has
// drained (checked lazily) - the copy never reads memory that has gone out of scope, whatever the runtime does with pageable sources.
// Host side of the small uploads: a ring of PINNED memory per context.  hipMemcpyAsync from pageable memory is staged by the runtime and
// holds the calling thread for ~10 us a piece; a flush of queued LoLa calls uploads a few hundred small tables.  A block of the ring is
// reused only after the stream has passed it: the ring synchronises once per lap (CN_PIN_RING_MIB, cn_get_option "pin_laps").
char *pin_block(cn_ctx *c, size_t bytes) {
    static const size_t cap = [] { const char *e = getenv("CN_PIN_RING_MIB"); const long m = e ? atol(e) : 0; return (size_t)(m >= 1 && m <= 1024 ? m : 32) << 20; }();    // 32 MiB: the unchanged CryptoNets caller uploads 0.4-0.75 MB of tables per prime and batch - a lap (one wait for the stream) every ~50 batches instead of every ~12 (8 MiB until round 6)
    bytes = (bytes + 63) & ~(size_t)63;
    if (bytes > cap / 2) return nullptr;
    if (!c->pin) { if (hipHostMalloc((void **)&c->pin, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); c->pin = nullptr; return nullptr; } c->pin_off = 0; }
    if (c->pin_off + bytes > cap) { c->pin_laps++; if (hipStreamSynchronize(c->stream) != hipSuccess || (c->stream2 && hipStreamSynchronize(c->stream2) != hipSuccess)) return nullptr; c->pin_off = 0; }
    char *p = c->pin + c->pin_off;
    c->pin_off += bytes;
    return p;
}
int upload_bytes(cn_ctx *c, const void *host, size_t bytes, void *dev) {
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
int place_table(cn_ctx *c, const void *host, size_t bytes, void *fallback, const void **dev) {
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
Buffer *getbuf(cn_ctx *c, cn_handle h, int kind) {
    Buffer *b = c->bufs.find(h);
    return (b && b->kind == kind) ? b : nullptr;
}
int range_ok(const Buffer *b, uint32_t first, uint32_t count, uint32_t stride) {
    if (!count) return 1;
    uint64_t last = (uint64_t)first + (uint64_t)(count - 1) * stride;
    return last < b->count;
}
// every entry point takes the context lock; all but the deferrable ones (cn_defer.hip) first drain the queue of deferred calls
// (the function bodies that start with LOCK / LOCK_ONLY run inside CnMutex::run - see API_BODY below: under the context lock, on the calling
// thread or, when the lock is held, on the holder's thread)
// Lock-free submission ("defer" = 2, cn_submit.h): every entry point that takes the lock first executes the records other threads have published up to
// this moment (ring_sync: in claim order, waiting for a slot that is claimed but not yet written); LOCK additionally reports the first error one of them ran into.

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
bool streams_share_a_queue(hipStream_t a, hipStream_t b) {
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
int pick_stream(cn_ctx *c) {
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
    { static std::atomic<uint64_t> next_uid{1}; c->uid = next_uid.fetch_add(1, std::memory_order_relaxed); }
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
int ctx_init(cn_ctx *c, uint32_t n, uint32_t k, int device, std::vector<uint64_t> &tw) {
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
    if (getenv("CN_SQ_HALVES")) c->sq_halves = std::max(0, std::min(2, atoi(getenv("CN_SQ_HALVES"))));
    if (getenv("CN_DEFER_STAGGER")) c->defer_stagger = atoi(getenv("CN_DEFER_STAGGER")) != 0;
    if (getenv("CN_ENC_FUSED")) c->enc_fused = atoi(getenv("CN_ENC_FUSED"));
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
void ctx_teardown(cn_ctx *ctx) {
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
    cn_stagger_forget(ctx);
    if (ctx->ev_front) (void)hipEventDestroy(ctx->ev_front);
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    cn_defer_delete(ctx->dq);
    delete ctx->ring; delete ctx->ready;
    delete ctx;
}
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
    if (!strcmp(name, "sq_halves")) { if (value < 0 || value > 2) return fail(CN_ERR_ARG, "sq_halves: 0, 1 or 2"); ctx->sq_halves = value; return 0; }
    if (!strcmp(name, "defer_stagger")) { ctx->defer_stagger = value != 0; return 0; }
    if (!strcmp(name, "enc_fused")) { ctx->enc_fused = value; return 0; }          // 0 three launches, 1 k_encrypt_fused, 2 k_encrypt_split
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
    else if (!strcmp(name, "pin_laps")) *value = (int)ctx->pin_laps;                      // laps of the pinned upload ring (each one waits for the stream)
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
    else if (!strcmp(name, "sq_halves")) *value = ctx->sq_halves;
    else if (!strcmp(name, "defer_stagger")) *value = ctx->defer_stagger;
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
bool keys_as_f64(const cn_ctx *ctx) { return ctx->use_f64 && ctx->hc.q_f64 && !ctx->legacy_ntt && ctx->hc.logn >= 10 && ctx->hc.logn <= 14; }
// coeff_form: the words are coefficient-form polynomials [..][k][N]; the device transforms them with its own tables (cn_load_key, form 1)
int set_key(cn_ctx *ctx, KsKey &slot, const uint64_t *words, size_t count, size_t expect, int is_dev, bool coeff_form) {
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
void pool_flush(cn_ctx *ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->pool) {
        std::vector<uint64_t *> keep;
        for (uint64_t *p : kv.second) { if (in_slab(ctx, p)) keep.push_back(p); else { (void)hipFree(p); ctx->pool_bytes -= kv.first; } }
        kv.second.swap(keep);
    }
}
int dev_alloc(cn_ctx *ctx, size_t bytes, uint64_t **out) {
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
int dev_release(cn_ctx *ctx, uint64_t *p, size_t bytes) {
    if (ctx->pool_bytes + bytes <= ctx->pool_max || ctx->capturing || in_slab(ctx, p)) { ctx->pool[bytes].push_back(p); ctx->pool_bytes += bytes; return 0; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(p));
    return 0;
}
int alloc_buf(cn_ctx *ctx, int kind, uint32_t count, uint32_t size, cn_handle *out) {
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
bool submit_async(const cn_ctx *ctx) { return ctx->defer.load(std::memory_order_relaxed) == 2 && !ctx->capturing.load(std::memory_order_relaxed); }
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
extern "C" int cn_free(cn_ctx *ctx, cn_handle h) {
    if (submit_async(ctx)) return ring_push(ctx, SUB_FREE, 1, h, 0, 0, 0, 0, 0, 0, 0);      // (a release must not overtake the published calls that read the array)
    API_BODY LOCK_ONLY; return free_body(ctx, h); API_END
}
int free_body(cn_ctx *ctx, cn_handle h) {
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
extern "C" int cn_free_many(cn_ctx *ctx, const cn_handle *h, uint32_t n) {
    if (submit_async(ctx) && h && n) {
        cn_handle *blk = (cn_handle *)malloc((size_t)n * sizeof(cn_handle));
        if (!blk) return fail(CN_ERR_ARG, "out of host memory");
        memcpy(blk, h, (size_t)n * sizeof(cn_handle));
        return ring_push(ctx, SUB_FREE_MANY, 1, 0, 0, 0, 0, 0, 0, n, (uint64_t)(uintptr_t)blk);
    }
    API_BODY LOCK_ONLY; return free_many_body(ctx, h, n); API_END
}
int free_many_body(cn_ctx *ctx, const cn_handle *h, uint32_t n) {
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
int free_graph(cn_ctx *ctx, Buffer &b) {
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
int ensure_index_map(cn_ctx *ctx) {
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

// ---------------------------------------------------------------- raw transforms / timing / stats
int raw_ntt(cn_ctx *ctx, void *p, uint32_t limbs, int base, int inverse) {
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

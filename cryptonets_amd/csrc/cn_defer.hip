// libcnhip.so host runtime (4/5): deferred submission of per-ciphertext calls (queue, hazard levels, flush as batched launches) and the consumer side of the
// lock-free submission ring (cn_submit.h).
#include "cn_api_shared.h"


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
DeferQueue *cn_defer_new() { return new DeferQueue(); }
void cn_defer_delete(DeferQueue *q) { delete q; }
bool cn_defer_pending(cn_ctx *ctx) { return ctx->dq && !ctx->dq->ops.empty(); }

int32_t defer_level(DeferQueue *q, const uint64_t *const *ins, uint32_t nin, const uint64_t *out) {
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
int defer_push(cn_ctx *ctx, DOp op, const uint64_t *const *ins, uint32_t nin) {
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
int flush_gemm_group(cn_ctx *ctx, DeferQueue *q, const std::vector<const DOp *> &ops, uint32_t K) {
    const uint32_t O = (uint32_t)ops.size();
    bool wsmall = true;
    for (const DOp *op : ops) wsmall = wsmall && gemm_weights_small(ctx, &q->wt[op->terms], K);
    const GemmArith ar = gemm_arith(ctx, wsmall);
    // the calls' lists, contiguous: A[o][kk] = address of term kk of output o (0 = padded tap), Wt[o][kk] its weight
    std::vector<uint64_t> A((size_t)O * K), Wt((size_t)O * K);
    for (uint32_t o = 0; o < O; o++) {
        memcpy(&A[(size_t)o * K], &q->addr[ops[o]->terms], (size_t)K * 8);
        memcpy(&Wt[(size_t)o * K], &q->wt[ops[o]->terms], (size_t)K * 8);
    }
    // Windows that share at least half of their inputs are merged in pairs, like cn_gemm_plan_create does for the batched path (pair_gather_lists: every shared input then
    // travels from L2 to a CU once - the CryptoNets convolution 371 -> 299 us).  Round 5 measured this inside the flush as a LOSS (16.1-17.7 against 15.1-15.5 ms per batch: the
    // pairing ran on the flushing thread under the lock every caller waited for).  Round 6, with the flush off the callers' path and index tables (below): alone on the device the paired
    // convolution is as fast as a plan's (310 against 500 us unpaired), end to end nothing moves and the pairing's host time sits at the head of a batch (profiles/r06_defer_pair_ab.txt).
    // OFF unless CN_DEFER_PAIR=1.
    static const bool pair_on = getenv("CN_DEFER_PAIR") && atoi(getenv("CN_DEFER_PAIR"));
    if (pair_on && ctx->gemm_pair && ar.small && K <= 64 && O >= 2) {
        std::unordered_map<uint64_t, int32_t> id_of; std::vector<uint64_t> addr_of;
        std::vector<int32_t> gidx((size_t)O * K);
        for (size_t x = 0; x < A.size(); x++) {
            if (!A[x]) { gidx[x] = -1; continue; }
            auto it = id_of.find(A[x]);
            if (it == id_of.end()) { it = id_of.emplace(A[x], (int32_t)addr_of.size()).first; addr_of.push_back(A[x]); }
            gidx[x] = it->second;
        }
        std::vector<uint64_t> W2; uint32_t K2 = K;
        if (pair_gather_lists(O, K2, gidx, Wt.data(), W2)) {
            K = K2; Wt.swap(W2);
            A.assign((size_t)O * K, 0);
            for (size_t x = 0; x < A.size(); x++) if (gidx[x] >= 0) A[x] = addr_of[gidx[x]];
        }
    }
    // group the outputs that gather the same inputs (PoolLayer: every map of one corner shares its patch; a dense layer: one group)
    std::map<std::vector<uint64_t>, std::vector<uint32_t>> groups;
    for (uint32_t o = 0; o < O; o++) groups[std::vector<uint64_t>(A.begin() + (size_t)o * K, A.begin() + (size_t)(o + 1) * K)].push_back(o);
    const uint32_t G = (uint32_t)groups.size();
    uint32_t M = 0;
    for (auto &g : groups) M = std::max<uint32_t>(M, (uint32_t)g.second.size());
    const bool mfma = gemm_mfma_ok(ctx, ar, M, K);
    const uint32_t Kp = mfma ? ((K + 31) / 32) * 32 : ((K + 15) & ~15u) + 16;       // gather rows: 16 spare entries (the VALU kernels request up to 2 x 8 terms ahead)
    const uint32_t NONE = 0xffffffffu;
    std::vector<uint64_t> hidx((size_t)G * Kp, 0), hoidx((size_t)G * M, 0), hbidx((size_t)G * M, 0);
    std::vector<uint32_t> member((size_t)G * M, NONE);
    bool any_bias = false, all_bias = true;
    const uint64_t *fallback = nullptr;
    {
        uint32_t g = 0;
        for (auto &kv : groups) {
            memcpy(&hidx[(size_t)g * Kp], kv.first.data(), (size_t)K * 8);
            for (uint32_t m = 0; m < kv.second.size(); m++) {
                const DOp *op = ops[kv.second[m]];
                member[(size_t)g * M + m] = kv.second[m]; hoidx[(size_t)g * M + m] = (uint64_t)op->out; hbidx[(size_t)g * M + m] = (uint64_t)op->bias;
                any_bias = any_bias || op->bias; all_bias = all_bias && op->bias;
            }
            for (uint64_t a : kv.first) if (a && !fallback) fallback = (const uint64_t *)a;
            g++;
        }
    }
    const bool small = ar.small, two = ar.two; const uint32_t lazy = ar.lazy; uint32_t MT = 1, WP = 0;
    bool one = false;
    std::vector<char> wbytes;
    auto row = [&](uint32_t g, uint32_t m) -> const uint64_t * { const uint32_t o = member[(size_t)g * M + m]; return o == NONE ? nullptr : &Wt[(size_t)o * K]; };
    if (mfma) {
        WP = gemm_weight_planes(ctx, Wt.data(), Wt.size());
        pack_gemm_mfma(ctx, G, M, K, WP, row, [&](uint32_t g, uint32_t kk) { return hidx[(size_t)g * Kp + kk] != 0; }, wbytes);
    } else {
        auto tap = [&](uint32_t g, uint32_t kk) { return hidx[(size_t)g * Kp + kk] != 0; };
        pack_gemm_weights(ctx, G, M, K, small, row, tap, MT, wbytes);
        one = gemm_one_limb(ctx, ar, G, M, K, row, tap);
    }
    // Index tables instead of address tables (round 6).  Every array the library hands out starts on a 256-byte boundary, so the operands of a flush group are 32-bit offsets in units
    // of 32 words from the lowest address among them - the INDEX-table kernels of a plan (one 16-byte scalar load per four gather entries, offsets multiplied on the scalar unit)
    // instead of the address-table variants (two loads per four, a 64-bit select per term): dense layer 242 -> 221 us alone on the device, convolution 506 -> 492
    // (310 with CN_DEFER_PAIR=1; profiles/r06_defer_pair_ab.txt); end to end neutral.  Needs a bias on every output or on none; CN_DEFER_REL=0 keeps the address tables (A/B).
    static const bool rel_on = !(getenv("CN_DEFER_REL") && !atoi(getenv("CN_DEFER_REL")));
    uint64_t ibase = ~0ull, obase_a = ~0ull, bbase = ~0ull;
    bool rel = rel_on && (!any_bias || all_bias);
    if (rel) {
        for (uint64_t a : hidx) if (a) ibase = std::min(ibase, a);
        for (uint64_t a : hoidx) if (a) obase_a = std::min(obase_a, a);
        for (uint64_t a : hbidx) if (a) bbase = std::min(bbase, a);
        auto fits = [](uint64_t a, uint64_t base) { return !a || (((a - base) & 255) == 0 && ((a - base) >> 8) < 0x7fffffffull); };
        for (uint64_t a : hidx) rel = rel && fits(a, ibase);
        for (uint64_t a : hoidx) rel = rel && fits(a, obase_a);
        for (uint64_t a : hbidx) rel = rel && fits(a, bbase);
        rel = rel && ibase != ~0ull && obase_a != ~0ull;
    }
    if (rel) {
        std::vector<int32_t> ridx(hidx.size(), -1), roidx(hoidx.size(), -1), rbidx(hbidx.size(), 0);
        for (size_t x = 0; x < hidx.size(); x++) if (hidx[x]) ridx[x] = (int32_t)((hidx[x] - ibase) >> 8);
        for (size_t x = 0; x < hoidx.size(); x++) if (hoidx[x]) roidx[x] = (int32_t)((hoidx[x] - obase_a) >> 8);
        if (any_bias) for (size_t x = 0; x < hbidx.size(); x++) if (hbidx[x]) rbidx[x] = (int32_t)((hbidx[x] - bbase) >> 8);
        const size_t off_oidx = al(ridx.size() * 4), off_bidx = off_oidx + al(roidx.size() * 4), off_w = off_bidx + al(rbidx.size() * 4);
        std::vector<char> host(off_w + al(wbytes.size()), 0);
        memcpy(host.data(), ridx.data(), ridx.size() * 4);
        memcpy(host.data() + off_oidx, roidx.data(), roidx.size() * 4);
        memcpy(host.data() + off_bidx, rbidx.data(), rbidx.size() * 4);
        memcpy(host.data() + off_w, wbytes.data(), wbytes.size());
        CHECK(ensure_scratch(ctx, al(host.size())));
        char *tables; CHECK(upload_tmp(ctx, host.data(), host.size(), &tables));
        GemmLaunch gl{small, two, false, MT, (const uint64_t *)ibase, tables, tables + off_w, tables + off_oidx, any_bias ? (const uint64_t *)bbase : nullptr, tables + off_bidx,
                      (uint64_t *)obase_a, G, M, K, lazy, Kp, 0, WP, (M + 31) / 32, (K + 31) / 32, 2, (uint32_t)ctx->gemm_order, one};
        gl.in_unit = gl.out_unit = gl.bias_unit = 32;
        return mfma ? cn_l_gemm_mfma(ctx, gl) : cn_l_gemm(ctx, gl);
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
int flush_elementwise_group(cn_ctx *ctx, const std::vector<const DOp *> &ops, int type) {
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
// Staggered plaintext-prime channels for the UNCHANGED caller (round 6, `defer_stagger`; THREE FORMS MEASURED, NONE WITH A GAIN - OFF BY DEFAULT, profiles/r06_stagger_ab.txt).  The reference runs the same layer on every plaintext prime from the same caller threads
// (EncryptedSealBfvVector.cs:225-236), so the two contexts of a CryptoNets device flush their big squaring layers within microseconds of each other and their kernel chains run in
// lock step - key switch beside key switch (both FP64-issue bound), element-wise BEHZ steps beside each other (both HBM bound).  The batched bench alternates the two channels'
// "fronts" (convolution + Multiply) with cn_ctx_wait_for, so that a front always runs beside the OTHER channel's key switch: 12.0-12.2 against 12.7-12.9 ms per batch, and it is the
// device-side ordering that does it, not the split into two calls (profiles/r06_stagger_ab.txt, visit N).  Here the library does the same for queued squaring groups of >= 256
// ciphertexts when exactly two contexts of a device take part: front #k of the first context (A) waits for front #(k-1) of the second (B), B's #k for A's #k - device-side events,
// strict alternation A0 B0 A1 B1 ... whatever order the host flushes them in.  When the partner has not ENQUEUED the front that is waited for yet (the two flushes are triggered
// microseconds apart by different threads), the flushing thread waits for it on the host, at most STAGGER_HOST_WAIT_US - a partner that never comes is skipped, counts that drift
// apart re-pair.  Two earlier forms measured no gain and are gone: waiting for whoever flushed last (a leader swap stalls the leader for the trailing context's front), and a fixed
// leader without the host wait (B's flush often precedes A's: no stagger, or a delayed front that is just a delay).  This third form engages (14 device-side waits in 16 fronts,
// one host wait ran out) and still measures 14.1-14.8 against 13.9 ms per batch (taps skipped, 16 threads, three alternating pairs): what the batched program gains from the same
// alternation does not carry over to the flush pattern of the unchanged caller (its convolution, dense and second squaring layers are flushed on their own, ahead of the fronts).
namespace {
struct StaggerPair {
    cn_ctx *c[2] = {nullptr, nullptr}; uint64_t n[2] = {0, 0}; hipEvent_t ev[2] = {nullptr, nullptr};
    bool off = false;                 // more than two contexts stagger on this device: nobody waits
    uint64_t waits = 0, timeouts = 0, ahead = 0, first = 0;      // CN_DEFER_TRACE: device-side waits placed / host waits that ran out / partner ahead / nothing to wait for
};
std::mutex g_front_mu;
StaggerPair g_pair[64];              // per device
}
static const uint32_t STAGGER_MIN_CTS = 256;
static const int STAGGER_HOST_WAIT_US = 400;
void cn_stagger_forget(cn_ctx *ctx) {             // a context that goes away takes its events with it: the pairing of its device starts over
    std::lock_guard<std::mutex> lk(g_front_mu);
    StaggerPair &p = g_pair[(unsigned)ctx->device % 64];
    if (p.c[0] == ctx || p.c[1] == ctx) {
        if (getenv("CN_DEFER_TRACE") && (p.waits || p.timeouts)) fprintf(stderr, "stagger device %d: %llu device-side waits, %llu host waits ran out, %llu partner ahead, %llu first, fronts %llu / %llu\n",
            ctx->device, (unsigned long long)p.waits, (unsigned long long)p.timeouts, (unsigned long long)p.ahead, (unsigned long long)p.first, (unsigned long long)p.n[0], (unsigned long long)p.n[1]);
        p = StaggerPair();
    }
}
static int stagger_slot(StaggerPair &p, cn_ctx *ctx) {      // (mutex held) 0 / 1, or -1: not a participant
    if (p.off) return -1;
    for (int i = 0; i < 2; i++) if (p.c[i] == ctx) return i;
    for (int i = 0; i < 2; i++) if (!p.c[i]) {        // a late joiner lines up with its partner's latest front: A's #k follows B's #(k-1), B's #k follows A's #k
        p.c[i] = ctx;
        p.n[i] = i == 0 ? p.n[1] : (p.n[0] ? p.n[0] - 1 : 0);
        return i;
    }
    p.off = true;
    return -1;
}
static int stagger_front_begin(cn_ctx *ctx) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        {
            std::lock_guard<std::mutex> lk(g_front_mu);
            StaggerPair &p = g_pair[(unsigned)ctx->device % 64];
            const int i = stagger_slot(p, ctx);
            if (i < 0 || !p.c[1 - i]) return 0;
            const uint64_t k = p.n[i], need = i == 0 ? k : k + 1, have = p.n[1 - i];       // A's #k follows B's #(k-1): B has recorded k fronts; B's #k follows A's #k: A has recorded k + 1
            if (need == 0) { p.first++; return 0; }
            if (have == need) { if (p.ev[1 - i]) HIPCHK(hipStreamWaitEvent(ctx->stream, p.ev[1 - i], 0)); p.waits++; return 0; }
            if (have > need) { p.ahead++; if (have > need + 2) { p.n[0] = p.n[1] = 0; } return 0; }   // the partner is ahead: nothing to wait for (far ahead: the counts drifted - re-pair)
            if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > STAGGER_HOST_WAIT_US) { p.timeouts++; return 0; }
        }
        if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > 4 * STAGGER_HOST_WAIT_US) return 0;
        std::this_thread::yield();                // the partner's flush is microseconds away (another thread, the other context's lock)
    }
}
static int stagger_front_end(cn_ctx *ctx) {
    if (!ctx->ev_front) HIPCHK(hipEventCreateWithFlags(&ctx->ev_front, hipEventDisableTiming));
    std::lock_guard<std::mutex> lk(g_front_mu);
    StaggerPair &p = g_pair[(unsigned)ctx->device % 64];
    const int i = stagger_slot(p, ctx);
    if (i < 0) return 0;
    HIPCHK(hipEventRecord(ctx->ev_front, ctx->stream));
    p.ev[i] = ctx->ev_front; p.n[i]++;
    static const bool trace = getenv("CN_DEFER_TRACE") != nullptr;
    if (trace && i == 1 && p.n[i] % 4 == 0) fprintf(stderr, "stagger device %d: fronts %llu / %llu, %llu device-side waits, %llu host waits ran out, %llu partner ahead, %llu first\n", ctx->device,
        (unsigned long long)p.n[0], (unsigned long long)p.n[1], (unsigned long long)p.waits, (unsigned long long)p.timeouts, (unsigned long long)p.ahead, (unsigned long long)p.first);
    return 0;
}
// all queued Multiply + Relinearize calls of one level: the batched BEHZ pipeline + ONE key switch, operands and results through tables
int flush_mulrelin_group(cn_ctx *ctx, const std::vector<const DOp *> &all) {
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
            const bool stagger = ctx->defer_stagger && c >= STAGGER_MIN_CTS && !ctx->capturing;
            auto mul = [&](uint32_t f, uint32_t n_) { return do_multiply(ctx, nullptr, 1, nullptr, 1, t3 + (size_t)f * 3 * kn, n_, da + f, (sq ? da : db) + f); };
            auto ksw = [&](uint32_t f, uint32_t n_) { uint64_t *t = t3 + (size_t)f * 3 * kn; return do_keyswitch(ctx, t + 2 * kn, 3 * kn, t, t + kn, 3 * kn, ctx->rlk, nullptr, n_, 0, nullptr, 0, dout + f); };
            if (ctx->sq_halves >= 2 && !stagger && !ctx->sq_overlap && ctx->hc.logn <= 13 && c >= SQ_HALVES_MIN && !ctx->capturing && aux_stream_ready(ctx)) { CHECK(pipelined_halves(ctx, c, mul, ksw)); continue; }
            if (stagger) CHECK(stagger_front_begin(ctx));
            CHECK(mul(0, c));
            if (stagger) CHECK(stagger_front_end(ctx));
            CHECK(ksw(0, c));
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
int ensure_stage(cn_ctx *ctx, size_t bytes) {
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
int copy_by_table(cn_ctx *ctx, const std::vector<Tab2> &tab, Tab2 *dtab, uint32_t words_per_item) {
    if (tab.empty()) return 0;
    const void *ptab;
    CHECK(place_table(ctx, tab.data(), tab.size() * sizeof(Tab2), dtab, &ptab));
    const uint32_t pairs = words_per_item / 2;
    hipLaunchKernelGGL(k_copy_tab, dim3((pairs + 255) / 256, (unsigned)tab.size()), dim3(256), 0, ctx->stream, (const Tab2 *)ptab, pairs);
    HIPCHK(hipGetLastError()); launch_count(ctx);
    return 0;
}
int flush_staged_group(cn_ctx *ctx, const std::vector<const DOp *> &all, int type) {
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
        // MultiplyPlain of ONE ciphertext by the plaintexts of several rows (the per-row DotProduct of a dense layer: every row multiplies the same vector): the broadcast form -
        // the ciphertext is transformed once, not once per row, and nothing is gathered (round 6)
        bool same_a = has_p && cnt >= 2;
        for (uint32_t i = 1; i < cnt && same_a; i++) same_a = ops[i]->a == ops[0]->a;
        for (uint32_t i = 0; i < cnt && same_a; i++) same_a = (const uint64_t *)ops[i]->out != ops[0]->a;      // (the shared operand is nobody's result)
        const bool da = same_a || spaced([](const DOp *o) { return o->a; }, ctw), db = has_b && spaced([](const DOp *o) { return o->b; }, ctw),
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
        case DOP_MULPLAIN: if (same_a) fa.count = 1; rc = mul_plain_impl(ctx, &fa, 0, same_a, &fp, 0, 1, &fo, 0, cnt); break;
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
int defer_staged(cn_ctx *ctx, int type, Buffer *A, uint32_t ai, Buffer *B, uint32_t bi, const uint64_t *plain, uint32_t pstride_words, Buffer *O, uint32_t oi,
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
int flush_encrypt_group(cn_ctx *ctx, const std::vector<const DOp *> &ops) {
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
bool zero_fold_ok(cn_ctx *ctx) {
    return ctx->pk && ctx->enc_fused && !ctx->legacy_ntt && ctx->use_f64 && ctx->hc.q_f64 && ctx->hc.logn >= 10 && ctx->hc.logn <= 13;
}
int flush_zero_folds(cn_ctx *ctx, DeferQueue *q, const std::vector<const DOp *> &gemms) {
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
    hipLaunchKernelGGL(k_sample_small, dim3((unsigned)(((uint64_t)cnt * (n / 16) + 255) / 256)), dim3(256), 0, ctx->stream, us, n, 0, 1u, cnt, key, 0ull, 0u, 0ull, (const EncTab *)dtab, cn_noise_table());
    hipLaunchKernelGGL(k_sample_small, dim3((unsigned)(((uint64_t)cnt * 2 * (n / 8) + 255) / 256)), dim3(256), 0, ctx->stream, es, n, 1, 2u, cnt, key, 0ull, 1u, 0ull, (const EncTab *)dtab, cn_noise_table());
    uint64_t qmax = 0; for (uint32_t j = 0; j < ctx->hc.k; j++) qmax = std::max(qmax, ctx->hc.q[j].q);
    if (!rr_ops[(qmax >> 44) ? POL_F64 : POL_F64L]->enc_fold(ctx, us, es, dfo, dft, (uint32_t)fo.size())) return fail(CN_ERR_ARG, "internal: zero-encryption fold without a kernel");
    HIPCHK(hipGetLastError()); launch_count(ctx, 3);
    ctx->st.ntt_forward_limbs += (uint64_t)fo.size() * ctx->hc.k; ctx->st.ntt_inverse_limbs += (uint64_t)fo.size() * 2 * ctx->hc.k;
    return 0;
}
int defer_encrypt(cn_ctx *ctx, const uint64_t *ptd, uint32_t pt_stride_words, Buffer *O, uint32_t oi, uint32_t count, uint64_t seed) {
    for (uint32_t c = 0; c < count; c++) {
        DOp op{DOP_ENCRYPT, 0, O->d + (size_t)(oi + c) * O->item_words, ptd ? ptd + (size_t)c * pt_stride_words : nullptr, nullptr, 0, 0, nullptr};
        op.nonce = seed; op.item = ctx->rng_item++;
        CHECK(defer_push(ctx, op, nullptr, 0));
    }
    return 0;
}

int cn_defer_flush(cn_ctx *ctx) {
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
bool deferring(cn_ctx *ctx) { return ctx->defer && !ctx->capturing; }

/* DenseMatrixBySparseVectorMultiply for ONE output block whose K input ciphertexts are separate objects (see include/cnhip.h) */
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
int scalar_dot_body(cn_ctx *ctx, const cn_handle *in, const uint32_t *in_idx, const uint64_t *w, uint32_t K, cn_handle out, uint32_t oi) {
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
int defer_addsub(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count, int op) {
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
int defer_add_plain(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, int subtract, cn_handle out, uint32_t oi, uint32_t count) {
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
int defer_mul_relin(cn_ctx *ctx, cn_handle a, uint32_t ai, uint32_t astride, cn_handle b, uint32_t bi, uint32_t bstride, cn_handle out, uint32_t oi, uint32_t count) {
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
void ready_refill(cn_ctx *ctx) {
    ReadyRing &r = *ctx->ready;
    if (ctx->capturing) return;
    if (r.misses.exchange(0, std::memory_order_relaxed)) r.target = std::min<uint32_t>(r.target * 2, (uint32_t)ReadyRing::CAP / 2);
    while (r.size() < r.target) {
        cn_handle h = 0;
        if (alloc_buf(ctx, 0, 1, 2, &h)) return;                 // out of memory: the callers fall back to the locked path and see the error there
        if (!r.push(h)) { Buffer *b = ctx->bufs.find(h); (void)dev_release(ctx, b->d, b->item_words * 8); ctx->bufs.erase(h); return; }
    }
}
int ring_exec(cn_ctx *ctx, const SubRec &r) {
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
void ring_drain(cn_ctx *ctx, uint64_t upto) {
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
int ring_sync(cn_ctx *ctx, bool report) {
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
int ring_push(cn_ctx *ctx, uint32_t type, uint32_t count, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t x, uint64_t arg) {
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

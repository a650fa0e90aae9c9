// libcnhip.so host runtime (2/5): the evaluator - linear operations, scalar GEMM planning, BEHZ multiply, key switching, rotations (launch logic behind the C ABI).
#include "cn_api_shared.h"

// ---------------------------------------------------------------- linear ops
int addsub(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count, int op) {
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
int addsub_body(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle b, uint32_t bi, cn_handle out, uint32_t oi, uint32_t count, int op) {
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
extern "C" int cn_add_plain(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, int subtract, cn_handle out, uint32_t oi, uint32_t count) {
    if (submit_async(ctx) && count <= DEFER_STAGED_MAX) return ring_push(ctx, SUB_ADD_PLAIN, count, a, ai, pt, pi, out, oi, subtract ? 1u : 0u, 0);
    API_BODY LOCK_ONLY; return add_plain_body(ctx, a, ai, pt, pi, subtract, out, oi, count); API_END
}
int add_plain_body(cn_ctx *ctx, cn_handle a, uint32_t ai, cn_handle pt, uint32_t pi, int subtract, cn_handle out, uint32_t oi, uint32_t count) {
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
int mul_plain_fused(cn_ctx *ctx, Buffer *A, uint32_t ai, bool a_bcast, Buffer *P, uint32_t pi, uint32_t pstride, Buffer *O, uint32_t oi, uint32_t count,
                           const BcastNext *nx) {
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
bool mul_plain_takes_bcast(cn_ctx *ctx, uint32_t count) { return ctx->mp_fused && ctx->mp_bcast && !ctx->legacy_ntt && ctx->hc.logn >= 10 && ctx->hc.logn <= 14 && count >= 4; }
int mul_plain_impl(cn_ctx *ctx, Buffer *A, uint32_t ai, bool a_bcast, Buffer *P, uint32_t pi, uint32_t pstride, Buffer *O, uint32_t oi, uint32_t count,
                          const BcastNext *nx) {
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
uint64_t lift_scalar(const DevConsts &hc, uint64_t w, uint32_t j) { return w >= hc.t_half ? w + hc.lift_inc[j] : w; }
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
// small signed weights (|w| < 2^20 after centring mod t - every PoolLayer weight round(w*scale) is): exact-FP64 limb-split kernel
bool gemm_weights_small(cn_ctx *ctx, const uint64_t *W, size_t count) {
    for (size_t x = 0; x < count; x++) {
        const uint64_t w = W[x], a = w >= ctx->hc.t_half ? ctx->hc.t.q - w : w;
        if (a >> 20) return false;
    }
    return true;
}
GemmArith gemm_arith(cn_ctx *ctx, bool weights_small) {
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

// ---- the matrix-core form of a scalar GEMM (k_scalar_gemm_mfma): eligibility, weight digit planes, A fragments
// 6 signed base-256 digits cover residues below 2^46 (x + 0x80..80 must stay below 2^48); i32 accumulators hold K * P * 2^14 < 2^31
bool gemm_mfma_ok(cn_ctx *ctx, const GemmArith &ar, uint32_t M, uint32_t K) {
    return ctx->gemm_mfma && ar.small && ar.bits <= 46 && M >= 16 && (uint64_t)K * 3 < (1u << 17) && !(ctx->hc.n & 31);
}
uint32_t gemm_weight_planes(cn_ctx *ctx, const uint64_t *W, size_t count) {
    uint64_t amax = 0;
    for (size_t x = 0; x < count; x++) { const uint64_t w = W[x]; amax = std::max(amax, w >= ctx->hc.t_half ? ctx->hc.t.q - w : w); }
    return amax <= 127 ? 1u : (amax <= 32639 ? 2u : 3u);          // signed digits -128..127: |w| <= 127 / 32639 / 8355711
}
int free_gemm_plan(cn_ctx *ctx, Buffer &b) {
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
bool pair_gather_lists(uint32_t O, uint32_t &K, std::vector<int32_t> &gidx, const uint64_t *W, std::vector<uint64_t> &W2) {
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
int build_gemm_plan(cn_ctx *ctx, const int32_t *idx, const uint64_t *W, uint32_t O, uint32_t K, Buffer *BP, cn_handle bias_pt, const int32_t *bias_idx,
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
int run_gemm_plan(cn_ctx *ctx, const GemmPlan &P, const char *tables, Buffer *I, Buffer *OB, uint32_t oi) {
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
bool run_intt_tensor(cn_ctx *c, const uint64_t *A, const uint64_t *B, uint64_t *D, uint32_t cnt, uint32_t base_off, uint32_t Lm) {
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
bool square_fused_ok(cn_ctx *c, uint32_t base_off, uint32_t Lm, bool &light) {
    if (!c->sq_fused || c->legacy_ntt || !c->use_f64 || c->hc.logn < 10 || c->hc.logn > 14) return false;
    light = true;
    for (uint32_t m = base_off; m < base_off + Lm; m++) {
        if (!c->hc.f64ok[m]) return false;
        uint64_t q = m < c->hc.k ? c->hc.q[m].q : c->hc.bsk[m - c->hc.k].q;
        if (q >> 44) light = false;
    }
    return true;
}
void run_square_fused(cn_ctx *c, const uint64_t *A, size_t astride, const uint64_t *const *atab, uint64_t *D, uint32_t cnt, uint32_t base_off, uint32_t Lm, bool light) {
    rr_ops[light ? POL_F64L : POL_F64]->square_fused(c, A, astride, atab, D, cnt, base_off, Lm);
    launch_count(c);
    c->st.ntt_forward_limbs += (uint64_t)cnt * 2 * Lm; c->st.ntt_inverse_limbs += (uint64_t)cnt * 3 * Lm;
}
// the context's second stream (squaring overlap): created on first use, kept only if it runs beside the context's own stream (a hardware queue of its own)
bool aux_stream_ready(cn_ctx *ctx) {
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
size_t mul_scratch_per_ct(cn_ctx *c, bool square) {
    size_t n = c->hc.n, k = c->hc.k, kb = c->hc.kb;
    size_t w = (square ? 1 : 2) * 2 * (k + kb) * n + 3 * (k + kb) * n;
    return al(w * 8) + 1024;
}
// a, b: pointers to first operand ciphertext (size 2); out3: [cnt][3][k][N]; scratch must be ensured by caller.  atab / btab: one
// operand address per ciphertext instead of a + ct*astride*ctw (deferred per-ciphertext calls; atab == btab: squarings)
int do_multiply(cn_ctx *ctx, const uint64_t *a, uint32_t astride, const uint64_t *b, uint32_t bstride, uint64_t *out3, uint32_t cnt,
                       const uint64_t *const *atab, const uint64_t *const *btab) {
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
int ensure_ks_part(cn_ctx *ctx, size_t need) {
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
uint32_t ks_digit_max_blocks() {              // (ciphertext, limb) blocks up to which the two-launch key switch runs one workgroup per DIGIT (above: per source limb); CN_KS_DIGIT_MAX overrides (A/B)
    // 10 since round 3 (1-2 ciphertexts; 32 before): with the four chains of an image on four hardware queues, per-source-limb workgroups cost the
    // chip less for 3-6 ciphertexts too (four chains 5.13-5.24 vs 5.36 ms per image, one chain alone unchanged; 50 / 65: 6.4 / 7.5 ms)
    static const uint32_t v = getenv("CN_KS_DIGIT_MAX") ? (uint32_t)atoi(getenv("CN_KS_DIGIT_MAX")) : 10u;
    return v;
}
uint32_t ks_wide_max_blocks() {               // (ciphertext, limb) blocks up to which a key switch runs as two launches; CN_KS_WIDE_MAX overrides (A/B)
    static const uint32_t v = getenv("CN_KS_WIDE_MAX") ? (uint32_t)atoi(getenv("CN_KS_WIDE_MAX")) : 160u;
    return v;
}
// the variant do_keyswitch takes for `cnt` ciphertexts: 0 = the fused kernel, 1 / 2 = two launches (KsArgs::mode)
int ks_planned_mode(cn_ctx *ctx, uint32_t cnt, int galois) {
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
int do_keyswitch(cn_ctx *ctx, const uint64_t *target, size_t tstride, const uint64_t *add0, const uint64_t *add1, size_t astride,
                        const KsKey &key, uint64_t *out, uint32_t cnt, int galois, const uint64_t *extra, size_t xstride,
                        uint64_t *const *out_tab, uint32_t perm_elt, const KsItem *items, uint32_t next_elt, uint64_t *next_out) {
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
bool ks_pair14_ok(cn_ctx *ctx, uint32_t cnt, int galois, const KsKey &key) {
    return ctx->ks_pair14 && ctx->ks_split14 && !ctx->legacy_ntt && ctx->hc.logn == 14 && ctx->hc.twdh && key.f64 && ks_planned_mode(ctx, cnt, galois) == 0;
}
uint32_t chunk_for(cn_ctx *ctx, size_t per_ct, uint32_t count) {
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
extern "C" int cn_mul_relin(cn_ctx *ctx, cn_handle a, uint32_t ai, uint32_t astride, cn_handle b, uint32_t bi, uint32_t bstride, cn_handle out,
                            uint32_t oi, uint32_t count) {
    if (submit_async(ctx) && count <= DEFER_STAGED_MAX) return ring_push(ctx, SUB_MUL_RELIN, count, a, ai, b, bi, out, oi, astride, bstride);      // PointwiseMultiply of one column
    API_BODY LOCK_ONLY; return mul_relin_body(ctx, a, ai, astride, b, bi, bstride, out, oi, count); API_END
}
int mul_relin_body(cn_ctx *ctx, cn_handle a, uint32_t ai, uint32_t astride, cn_handle b, uint32_t bi, uint32_t bstride, cn_handle out, uint32_t oi, uint32_t count) {
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
        auto mul = [&](uint32_t f, uint32_t n_) { return do_multiply(ctx, pa + (size_t)(s + f) * astride * A->item_words, astride, pb + (size_t)(s + f) * bstride * B->item_words, bstride, t3 + (size_t)f * 3 * kn, n_); };
        auto ksw = [&](uint32_t f, uint32_t n_) { uint64_t *t = t3 + (size_t)f * 3 * kn; return do_keyswitch(ctx, t + 2 * kn, 3 * kn, t, t + kn, 3 * kn, ctx->rlk, O->d + (oi + s + f) * O->item_words, n_, 0); };
        if (ctx->sq_halves && !ctx->sq_overlap && ctx->hc.logn <= 13 && c >= SQ_HALVES_MIN && !ctx->capturing && aux_stream_ready(ctx)) { CHECK(pipelined_halves(ctx, c, mul, ksw)); continue; }
        CHECK(mul(0, c));
        CHECK(ksw(0, c));
    }
    ctx->st.Relinarization += count;
    return 0;
}

// ---------------------------------------------------------------- rotations
// in/out device pointers to size-2 ciphertext arrays; tmp holds count size-2 ciphertexts
// acc != nullptr: out = acc + galois(in) in the same launches (acc may alias out and/or in)
// pre: sigma_elt(c1) of `in` if somebody has produced it already ([ct][k][N]); next_elt / next_out: see do_keyswitch (both only on the one-launch N = 16384 path)
int do_galois(cn_ctx *ctx, const uint64_t *in, uint64_t elt, uint64_t *out, uint64_t *tmp, uint32_t count, const uint64_t *acc,
                     const uint64_t *pre, uint64_t next_elt, uint64_t *next_out) {
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
bool shifted_overlap(const Buffer *I, uint32_t ii, const Buffer *O, uint32_t oi, uint32_t count) { return I == O && ii != oi && ii < oi + count && oi < ii + count; }
int galois_impl(cn_ctx *ctx, Buffer *I, uint32_t ii, uint64_t elt, Buffer *O, uint32_t oi, uint32_t count) {
    if (!range_ok(I, ii, count) || !range_ok(O, oi, count)) return fail(CN_ERR_ARG, "index out of range");
    if (!count) return 0;
    CHECK(ensure_scratch(ctx, al(count * ctx->ctw2 * 8)));
    uint64_t *tmp = salloc<uint64_t>(ctx, count * ctx->ctw2);
    return do_galois(ctx, I->d + ii * I->item_words, elt, O->d + oi * O->item_words, tmp, count);
}
bool galois_key_present(cn_ctx *ctx, uint64_t elt) { auto it = ctx->gk.find(elt); return it != ctx->gk.end() && it->second.d; }
extern "C" int cn_apply_galois(cn_ctx *ctx, cn_handle in, uint32_t ii, uint64_t elt, cn_handle out, uint32_t oi, uint32_t count) { API_BODY
    LOCK; GETCT(I, in, 2); GETCT(O, out, 2);
    return galois_impl(ctx, I, ii, elt, O, oi, count);
API_END }
// Evaluator::rotate_internal: direct key if present, otherwise non-adjacent-form decomposition
bool has_direct_key(cn_ctx *ctx, int steps) {
    uint64_t elt = cn_galois_elt_from_step(ctx, steps);
    auto it = ctx->gk.find(elt);
    return elt && it != ctx->gk.end() && it->second.d;
}
int rotate_rec(cn_ctx *ctx, uint64_t *cur, int steps, uint64_t *tmp, uint32_t count) {
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
int rotate_rows_impl(cn_ctx *ctx, Buffer *I, uint32_t ii, int steps, Buffer *O, uint32_t oi, uint32_t count) {
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
int rotate_check(cn_ctx *ctx, int steps) {
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
int rotation_hops(cn_ctx *ctx, int steps, std::vector<uint64_t> &elts) {
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
int rotate_jobs(cn_ctx *ctx, std::vector<RotJob> &jobs) {
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
int rotate_rows_add_impl(cn_ctx *ctx, Buffer *I, uint32_t ii, int steps, Buffer *A, uint32_t ai, Buffer *O, uint32_t oi, uint32_t count) {
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
int rotate_columns_add_impl(cn_ctx *ctx, Buffer *I, uint32_t ii, Buffer *A, uint32_t ai, Buffer *O, uint32_t oi, uint32_t count) {
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
std::vector<uint64_t> sum_slots_chain_elts(cn_ctx *ctx, uint32_t count, uint32_t length) {
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
int sum_slots_impl(cn_ctx *ctx, Buffer *H, uint32_t first, uint32_t count, uint32_t length, bool first_ready) {
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

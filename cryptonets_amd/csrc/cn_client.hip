// libcnhip.so host runtime (3/5): the data owner's side on the device - keys, ChaCha20 sampler, keygen, encrypt, decrypt, noise.
#include "cn_api_shared.h"

// ---------------------------------------------------------------- client side on the device
int set_plain_key(cn_ctx *ctx, uint64_t **slot, const uint64_t *words, size_t count, size_t expect, bool is_dev, bool coeff_form) {
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
RngKey rng_key_of(const cn_ctx *ctx) { RngKey k; memcpy(k.k, ctx->rng_key, sizeof k.k); return k; }
// thresholds of sample_noise8 (cn_dev_common.hip.h): cumulative distribution of |x|, x ~ N(0, 3.2^2) conditioned on |x| <= 19.2 (SEAL 3.2: noise_standard_deviation 3.20, noise_max_deviation 6 sigma)
const NoiseTab &cn_noise_table() {
    static const NoiseTab tab = [] {
        NoiseTab t;
        const long double sigma = 3.2L, root2 = 1.41421356237309504880168872420969808L, norm = erfl(19.2L / (sigma * root2));
        for (int i = 0; i < 19; i++) {
            const long double c = erfl((long double)(i + 1) / (sigma * root2)) / norm;             // P(|x| < i + 1 | clipped)
            t.thr[i] = c >= 1.0L ? 0x7fffffffffffffffull : (uint64_t)floorl(c * 9223372036854775808.0L);
        }
        return t;
    }();
    return tab;
}
// `polys` polynomials [polys][k][N] of residues: kind 0 ternary, 1 clipped normal (both drawn ONCE per coefficient into an int8 array in
// scratch - the caller's ensure_scratch leaves room for polys * N bytes - and expanded to the k limbs), 2 uniform per limb
int sample_poly(cn_ctx *ctx, uint64_t *dst, uint32_t polys, int kind, uint64_t seed, uint64_t stream) {
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
                           ctx->rng_item, (const EncTab *)nullptr, cn_noise_table());
        hipLaunchKernelGGL(k_expand_small, dim3(polys * k * ctx->chunks), dim3(ctx->bs), 0, ctx->stream, small, dst, ctx->dc, ctx->chunks);
        launch_count(ctx);
    }
    HIPCHK(hipGetLastError()); launch_count(ctx);
    ctx->rng_item += polys;
    return 0;
}
// one key-switch key for the NTT-form target polynomial snew: [(l,d)][2][k][N]
int gen_ksk(cn_ctx *ctx, const uint64_t *snew, int dbc, const uint32_t *dig, uint32_t tot, uint64_t seed, uint64_t *key, uint64_t *e) {
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
int adopt_ksk(cn_ctx *ctx, KsKey &slot, uint64_t *dev, size_t words) {          // takes ownership of a device buffer
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
int encrypt_chain(cn_ctx *ctx, uint32_t cnt, const uint64_t *ptd, uint32_t pt_stride_words, uint64_t *out, uint64_t seed, const EncTab *htab) {
    const uint32_t n = ctx->hc.n, k = ctx->hc.k; const size_t kn = (size_t)k * n;
    CHECK(ensure_scratch(ctx, al((size_t)cnt * kn * 8) + al((size_t)cnt * n) + al((size_t)cnt * 2 * n) + (htab ? al(cnt * sizeof(EncTab)) : 0) + 1024));
    uint64_t *u = salloc<uint64_t>(ctx, (size_t)cnt * kn);
    int8_t *us = salloc<int8_t>(ctx, (size_t)cnt * n), *es = salloc<int8_t>(ctx, (size_t)cnt * 2 * n);
    EncTab *dtab = nullptr;
    if (htab) CHECK(upload_tmp(ctx, htab, cnt, &dtab));
    if (!u || !us || !es) return fail(CN_ERR_HIP, "internal: scratch exhausted in encrypt");
    const RngKey key = rng_key_of(ctx);
    const uint64_t item0 = ctx->rng_item;
    hipLaunchKernelGGL(k_sample_small, dim3((unsigned)(((uint64_t)cnt * (n / 16) + 255) / 256)), dim3(256), 0, ctx->stream, us, n, 0, 1u, cnt, key, seed, 0u, item0, (const EncTab *)dtab, cn_noise_table());
    hipLaunchKernelGGL(k_sample_small, dim3((unsigned)(((uint64_t)cnt * 2 * (n / 8) + 255) / 256)), dim3(256), 0, ctx->stream, es, n, 1, 2u, cnt, key, seed, 1u, item0, (const EncTab *)dtab, cn_noise_table());
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
extern "C" int cn_encrypt(cn_ctx *ctx, cn_handle pt, uint32_t pi, uint32_t pt_stride, cn_handle out, uint32_t oi, uint32_t count, uint64_t seed) {
    if (submit_async(ctx) && count && count <= 4) return ring_push(ctx, SUB_ENCRYPT, count, 0, 0, pt, pi, out, oi, pt_stride, seed);
    API_BODY LOCK_ONLY; return encrypt_body(ctx, pt, pi, pt_stride, out, oi, count, seed); API_END
}
int encrypt_body(cn_ctx *ctx, cn_handle pt, uint32_t pi, uint32_t pt_stride, cn_handle out, uint32_t oi, uint32_t count, uint64_t seed) {
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
template <int K> static void launch_dec_scale(cn_ctx *c, const uint64_t *c0, size_t stride, const uint64_t *acc, uint64_t *plain, uint32_t cnt) {
    hipLaunchKernelGGL(k_decrypt_scale<K>, dim3(cnt * c->chunks), dim3(c->bs), 0, c->stream, c0, stride, acc, plain, c->dc, c->chunks);
}
// acc[ct][j] <- c1 s (+ c2 s^2) in coefficient form: the part of the decryption phase that needs the secret key
int decrypt_phase(cn_ctx *ctx, Buffer *I, uint32_t ci, uint32_t count, uint64_t *&acc) {
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

// Launchers of the register-radix transform kernels for ONE arithmetic policy (RR_POLICY, RR_NAME): included by cn_l_rr_u64.hip,
// cn_l_rr_f64.hip and cn_l_rr_f64l.hip, so that the three policies compile in parallel.
#include "cn_runtime.h"
#include "cn_k_rr.hip.h"
#include <algorithm>

typedef RR_POLICY AR;
static constexpr bool kF64 = std::is_same<typename AR::T, double>::value;

template <class K> static int big_lds(K kern, size_t bytes) {
    HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
template <int L> static int set_attrs_l(size_t bytes) {
    CHECK(big_lds(k_ntt_rr<L, AR, false>, bytes)); CHECK(big_lds(k_ntt_rr<L, AR, true>, bytes)); CHECK(big_lds(k_intt_tensor<L, AR>, bytes));
    if constexpr (kF64) {
        CHECK(big_lds(k_square_fused<L, AR>, bytes));
        if constexpr (L <= 13) { CHECK(big_lds(k_square_fused<L, AR, true>, bytes + ((size_t)8 << L))); CHECK(big_lds(k_square_pipe<L, AR>, bytes + ((size_t)8 << L))); }
    }
    CHECK(big_lds(k_lift_ntt<L, AR>, bytes)); CHECK(big_lds(k_mul_plain_fused<L, AR>, bytes)); CHECK(big_lds(k_mul_plain_bcast<L, AR>, bytes));
#ifdef RR_ENC_TAIL
    CHECK(big_lds(k_encrypt_tail<L, AR>, bytes));
#endif
    if constexpr (L <= 13) { CHECK(big_lds(k_encrypt_fused<L, AR>, bytes)); CHECK(big_lds(k_encrypt_split<L, AR>, bytes)); }
    if constexpr (L <= 13 && kF64) CHECK(big_lds(k_encrypt_fold<L, AR>, bytes));
    return 0;
}
static int set_attrs(uint32_t logn, size_t bytes) {      // transforms whose padded LDS image exceeds the default dynamic-LDS limit (N >= 8192)
    if constexpr (kF64) { if (logn == 12) { CHECK(big_lds(k_square_fused<12, AR, true>, bytes + ((size_t)8 << 12))); CHECK(big_lds(k_square_pipe<12, AR>, bytes + ((size_t)8 << 12))); } }   // image + parked operand / root table = 66.5 KiB
    if (logn == 13) return set_attrs_l<13>(bytes);
    if (logn == 14) return set_attrs_l<14>(bytes);
    return 0;
}
#define BY_SIZE(fn, ...) switch (c->hc.logn) { \
    case 10: fn<10>(__VA_ARGS__); return true; case 11: fn<11>(__VA_ARGS__); return true; case 12: fn<12>(__VA_ARGS__); return true; \
    case 13: fn<13>(__VA_ARGS__); return true; case 14: fn<14>(__VA_ARGS__); return true; default: return false; }

template <int L> static void l_ntt(cn_ctx *c, uint64_t *data, uint32_t limbs, uint32_t base_off, uint32_t nmod, int inverse) {
    const size_t lds = (size_t)ntt_lds_words(1u << L) * 8;
    if (inverse) hipLaunchKernelGGL((k_ntt_rr<L, AR, true>), dim3(limbs), dim3(NttPlan<L>::NT), lds, c->stream, data, c->dc, base_off, nmod);
    else hipLaunchKernelGGL((k_ntt_rr<L, AR, false>), dim3(limbs), dim3(NttPlan<L>::NT), lds, c->stream, data, c->dc, base_off, nmod);
}
static bool ntt(cn_ctx *c, uint64_t *data, uint32_t limbs, uint32_t base_off, uint32_t nmod, int inverse) { BY_SIZE(l_ntt, c, data, limbs, base_off, nmod, inverse) }

// tensor product fused into the inverse transform
template <int L> static void l_intt_tensor(cn_ctx *c, const uint64_t *A, const uint64_t *B, uint64_t *D, uint32_t cnt, uint32_t base_off, uint32_t Lm) {
    hipLaunchKernelGGL((k_intt_tensor<L, AR>), dim3(cnt * 3 * Lm), dim3(NttPlan<L>::NT), (size_t)ntt_lds_words(1u << L) * 8, c->stream, A, B, D, c->dc, base_off, Lm);
}
static bool intt_tensor(cn_ctx *c, const uint64_t *A, const uint64_t *B, uint64_t *D, uint32_t cnt, uint32_t base_off, uint32_t Lm) { BY_SIZE(l_intt_tensor, c, A, B, D, cnt, base_off, Lm) }

// squaring: forward transforms, tensor and inverse transforms of one (ciphertext, limb) in ONE kernel (FP64 policies)
template <int L> static void l_square_fused(cn_ctx *c, const uint64_t *A, size_t astride, const uint64_t *const *atab, uint64_t *D, uint32_t cnt, uint32_t base_off, uint32_t Lm) {
    if constexpr (kF64) {
        const size_t lds = (size_t)ntt_lds_words(1u << L) * 8;
        if constexpr (L <= 13) {
            // pipelined resident kernel (k_square_pipe): one workgroup per CU and modulus for the whole launch; pays once a workgroup squares several blocks
            const uint32_t per_limb = std::min<uint32_t>((uint32_t)std::max(1, c->cus) / Lm, cnt);
            if (per_limb >= 1 && (c->sq_pipe == 2 || (c->sq_pipe && cnt >= 4 * per_limb))) {
                hipLaunchKernelGGL((k_square_pipe<L, AR>), dim3(per_limb * Lm), dim3(NttPlan<L>::NT), lds + ((size_t)8 << L), c->stream, A, astride, atab, D, c->dc, base_off, Lm, cnt);
                return;
            }
            if (c->sq_lds) {             // NTT-form operand parked in LDS (one workgroup per CU) instead of in the outputs' place (two)
                hipLaunchKernelGGL((k_square_fused<L, AR, true>), dim3(cnt * Lm), dim3(NttPlan<L>::NT), lds + ((size_t)8 << L), c->stream, A, astride, atab, D, c->dc, base_off, Lm);
                return;
            }
        }
        hipLaunchKernelGGL((k_square_fused<L, AR>), dim3(cnt * Lm), dim3(NttPlan<L>::NT), lds, c->stream, A, astride, atab, D, c->dc, base_off, Lm);
    }
}
static bool square_fused(cn_ctx *c, const uint64_t *A, size_t astride, const uint64_t *const *atab, uint64_t *D, uint32_t cnt, uint32_t base_off, uint32_t Lm) {
    if (!kF64) return false;
    BY_SIZE(l_square_fused, c, A, astride, atab, D, cnt, base_off, Lm)
}

// dense MultiplyPlain in two launches (k_lift_ntt, k_mul_plain_fused)
template <int L> static void l_mul_plain_fused(cn_ctx *c, const uint64_t *pt, uint32_t pitch, uint32_t npt, uint64_t *lift, const uint64_t *src, size_t sstride,
                                               uint32_t pstride, uint64_t *out, uint32_t count, uint32_t polys) {
    const size_t lds = (size_t)ntt_lds_words(1u << L) * 8;
    hipLaunchKernelGGL((k_lift_ntt<L, AR>), dim3(npt * c->hc.k), dim3(NttPlan<L>::NT), lds, c->stream, pt, pitch, lift, c->dc);
    hipLaunchKernelGGL((k_mul_plain_fused<L, AR>), dim3(count * polys * c->hc.k), dim3(NttPlan<L>::NT), lds, c->stream, src, sstride, lift, pstride, out, c->dc, polys);
}
static bool mul_plain_fused(cn_ctx *c, const uint64_t *pt, uint32_t pitch, uint32_t npt, uint64_t *lift, const uint64_t *src, size_t sstride, uint32_t pstride,
                            uint64_t *out, uint32_t count, uint32_t polys) { BY_SIZE(l_mul_plain_fused, c, pt, pitch, npt, lift, src, sstride, pstride, out, count, polys) }

// one ciphertext (NTT form in ctn) times `count` plaintexts: one launch
template <int L> static void l_mul_plain_bcast(cn_ctx *c, const uint64_t *pt, uint32_t pitch, const uint64_t *ctn, uint64_t *out, uint32_t count, uint32_t polys, uint32_t next_elt,
                                               uint64_t *next_out) {
    hipLaunchKernelGGL((k_mul_plain_bcast<L, AR>), dim3(count * polys * c->hc.k), dim3(NttPlan<L>::NT), (size_t)ntt_lds_words(1u << L) * 8, c->stream, pt, pitch, ctn, out, c->dc, polys,
                       next_elt, next_out);
}
static bool mul_plain_bcast(cn_ctx *c, const uint64_t *pt, uint32_t pitch, const uint64_t *ctn, uint64_t *out, uint32_t count, uint32_t polys, uint32_t next_elt, uint64_t *next_out) {
    BY_SIZE(l_mul_plain_bcast, c, pt, pitch, ctn, out, count, polys, next_elt, next_out)
}

#ifdef RR_ENC_TAIL
template <int L> static void l_enc_tail(cn_ctx *c, const uint64_t *u, const uint64_t *pt, uint32_t pts, uint64_t *out, uint32_t cnt, const int8_t *noise, const void *tab) {
    hipLaunchKernelGGL((k_encrypt_tail<L, AR>), dim3(cnt * 2 * c->hc.k), dim3(NttPlan<L>::NT), (size_t)ntt_lds_words(1u << L) * 8, c->stream, u, c->pk, pt, pts, out, c->dc,
                       noise, (const EncTab *)tab);
}
static bool enc_tail(cn_ctx *c, const uint64_t *u, const uint64_t *pt, uint32_t pts, uint64_t *out, uint32_t cnt, const int8_t *noise, const void *tab) {
    BY_SIZE(l_enc_tail, c, u, pt, pts, out, cnt, noise, tab)
}
#else
static bool enc_tail(cn_ctx *, const uint64_t *, const uint64_t *, uint32_t, uint64_t *, uint32_t, const int8_t *, const void *) { return false; }
#endif

// Encryptor.Encrypt behind the samplers as ONE kernel (N <= 8192; N = 16384 keeps the three-launch chain: 1024-thread workgroups have 128 VGPRs per thread)
template <int L> static void l_enc_fused(cn_ctx *c, const int8_t *us, const uint64_t *pt, uint32_t pts, uint64_t *out, uint32_t cnt, const int8_t *noise, const void *tab) {
    if constexpr (L <= 13) {
        if (c->enc_fused == 2 && kF64)               // (the integer policy spills 17-21 registers at 128: it keeps the one-block form)
            hipLaunchKernelGGL((k_encrypt_split<L, AR>), dim3(cnt * 2 * c->hc.k), dim3(NttPlan<L>::NT), (size_t)ntt_lds_words(1u << L) * 8, c->stream, us, c->pk, pt, pts, out, c->dc,
                               noise, (const EncTab *)tab);
        else
            hipLaunchKernelGGL((k_encrypt_fused<L, AR>), dim3(cnt * c->hc.k), dim3(NttPlan<L>::NT), (size_t)ntt_lds_words(1u << L) * 8, c->stream, us, c->pk, pt, pts, out, c->dc,
                               noise, (const EncTab *)tab);
    }
}
static bool enc_fused(cn_ctx *c, const int8_t *us, const uint64_t *pt, uint32_t pts, uint64_t *out, uint32_t cnt, const int8_t *noise, const void *tab) {
    if (c->hc.logn > 13) return false;
    BY_SIZE(l_enc_fused, c, us, pt, pts, out, cnt, noise, tab)
}
// weighted sums of fresh zero encryptions folded onto scalar-product outputs (k_encrypt_fold; FP64 policies, N <= 8192)
template <int L> static void l_enc_fold(cn_ctx *c, const int8_t *us, const int8_t *noise, const void *fout, const void *terms, uint32_t outputs) {
    if constexpr (L <= 13 && kF64)
        hipLaunchKernelGGL((k_encrypt_fold<L, AR>), dim3(outputs * 2 * c->hc.k), dim3(NttPlan<L>::NT), (size_t)ntt_lds_words(1u << L) * 8, c->stream, us, c->pk, c->dc, noise,
                           (const FoldOut *)fout, (const FoldTerm *)terms);
}
static bool enc_fold(cn_ctx *c, const int8_t *us, const int8_t *noise, const void *fout, const void *terms, uint32_t outputs) {
    if (!kF64 || c->hc.logn > 13) return false;
    BY_SIZE(l_enc_fold, c, us, noise, fout, terms, outputs)
}
#ifndef __HIP_DEVICE_COMPILE__      // host-side table (in the device pass a const global would be emitted as device data)
extern const RrOps RR_NAME = {set_attrs, ntt, intt_tensor, square_fused, mul_plain_fused, enc_tail, enc_fused, mul_plain_bcast, enc_fold};
#endif

// HIP kernels of libcnhip.so (gfx950 / CDNA4, wave64).  Integer VALU only: all arithmetic is
// unsigned 64-bit modular arithmetic over RNS residues (no MFMA - nothing here is a float contraction).
//
// Data layout in HBM: ciphertext array = [ct][poly][limb][N] u64 (SEAL's per-ciphertext layout,
// contiguous over the batch), so lane i of a wave touches coefficient i of one limb: every global
// access below is a fully coalesced 8 B/lane (512 B/wave) stream.
#pragma once
#include <hip/hip_runtime.h>
#include "cn_internal.h"
#include "cn_ntt_core.hip.h"
#include <type_traits>

typedef unsigned __int128 u128;
#define DEV __device__ __forceinline__

// ------------------------------------------------------------------ modular helpers
DEV uint64_t bred128(uint64_t x0, uint64_t x1, const DMod &m) {   // Barrett, x = x1:x0 < 2^128
    uint64_t carry = __umul64hi(x0, m.r0);
    uint64_t t2lo = x0 * m.r1, t2hi = __umul64hi(x0, m.r1);
    uint64_t tmp1 = t2lo + carry, tmp3 = t2hi + (tmp1 < carry);
    uint64_t t3lo = x1 * m.r0, t3hi = __umul64hi(x1, m.r0);
    uint64_t s = tmp1 + t3lo;
    carry = t3hi + (s < tmp1);
    uint64_t qhat = x1 * m.r1 + tmp3 + carry;
    uint64_t r = x0 - qhat * m.q;
    return r >= m.q ? r - m.q : r;
}
DEV uint64_t bred128(u128 x, const DMod &m) { return bred128((uint64_t)x, (uint64_t)(x >> 64), m); }
DEV uint64_t mulmod(uint64_t a, uint64_t b, const DMod &m) { return bred128(a * b, __umul64hi(a, b), m); }
DEV uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) { uint64_t s = a + b; return s >= q ? s - q : s; }
DEV uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }
DEV uint64_t negmod(uint64_t a, uint64_t q) { return a ? q - a : 0; }
// Harvey/Shoup lazy product: y*w mod q in [0,2q) for any 64-bit y, ws = floor(w*2^64/q)
DEV uint64_t shoup_lazy(uint64_t y, uint64_t w, uint64_t ws, uint64_t q) { return y * w - __umul64hi(ws, y) * q; }

// ------------------------------------------------------------------ in-LDS negacyclic NTT
// One workgroup owns one limb (N words in LDS). Cooley-Tukey DIT, bit-reversed twiddle table,
// values kept lazily in [0,4q); output order = SEAL's (bit-reversed evaluations).
DEV void ntt_fwd_lds(uint64_t *s, const uint64_t *__restrict__ w, const uint64_t *__restrict__ ws, uint64_t q, uint32_t n) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x, half = n >> 1;
    const uint64_t q2 = 2 * q;
    uint32_t logt = 31 - __clz(half);
    for (uint32_t m = 1; m < n; m <<= 1, logt--) {
        const uint32_t t = 1u << logt;
        for (uint32_t b = tid; b < half; b += nt) {
            uint32_t i = b >> logt, j = b & (t - 1);
            uint32_t ia = (i << (logt + 1)) + j, ib = ia + t;
            uint64_t W = w[m + i], Ws = ws[m + i];
            uint64_t X = s[ia], Y = s[ib];
            X -= (X >= q2) ? q2 : 0;
            uint64_t Q = shoup_lazy(Y, W, Ws, q);
            s[ia] = X + Q;
            s[ib] = X + q2 - Q;
        }
        __syncthreads();
    }
}
// Gentleman-Sande inverse; input canonical or in [0,2q), output in [0,2q) WITHOUT the 1/N factor.
DEV void ntt_inv_lds(uint64_t *s, const uint64_t *__restrict__ iw, const uint64_t *__restrict__ iws, uint64_t q, uint32_t n) {
    const uint32_t tid = threadIdx.x, nt = blockDim.x, half = n >> 1;
    const uint64_t q2 = 2 * q;
    uint32_t logt = 0;
    for (uint32_t m = half; m >= 1; m >>= 1, logt++) {
        const uint32_t t = 1u << logt;
        for (uint32_t b = tid; b < half; b += nt) {
            uint32_t i = b >> logt, j = b & (t - 1);
            uint32_t ia = (i << (logt + 1)) + j, ib = ia + t;
            uint64_t W = iw[m + i], Ws = iws[m + i];
            uint64_t U = s[ia], V = s[ib];
            uint64_t S = U + V;
            S -= (S >= q2) ? q2 : 0;
            s[ia] = S;
            s[ib] = shoup_lazy(U + q2 - V, W, Ws, q);
        }
        __syncthreads();
    }
}
DEV uint64_t canon4(uint64_t v, uint64_t q) { uint64_t q2 = 2 * q; v -= (v >= q2) ? q2 : 0; v -= (v >= q) ? q : 0; return v; }

// tw layout: modulus m -> tw + m*4n : w, ws, iw, iws
DEV const uint64_t *tw_of(const DevConsts *C, uint32_t mod) { return C->tw + (size_t)mod * 4 * C->n; }

// batched in-place NTT: block b transforms limb b; modulus = base_off + (b % nmod)
__global__ void __launch_bounds__(1024) k_ntt(uint64_t *data, const DevConsts *__restrict__ C, uint32_t base_off, uint32_t nmod, int inverse) {
    extern __shared__ uint64_t s[];
    const uint32_t n = C->n, mod = base_off + blockIdx.x % nmod;
    const uint64_t q = mod < C->k ? C->q[mod].q : (mod < C->k + C->kb ? C->bsk[mod - C->k].q : C->t.q);
    uint64_t *x = data + (size_t)blockIdx.x * n;
    const uint64_t *tw = tw_of(C, mod);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s[i] = x[i];
    __syncthreads();
    if (!inverse) {
        ntt_fwd_lds(s, tw, tw + n, q, n);
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) x[i] = canon4(s[i], q);
    } else {
        ntt_inv_lds(s, tw + 2 * (size_t)n, tw + 3 * (size_t)n, q, n);
        const uint64_t ni = C->ninv[mod], nis = C->ninvs[mod];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { uint64_t v = shoup_lazy(s[i], ni, nis, q); x[i] = v >= q ? v - q : v; }
    }
}

// ------------------------------------------------------------------ element-wise kernels
// grid.x = limbs * chunks ; limb index is block-uniform so moduli come from scalar loads.
struct Geo { uint32_t chunks, bs; };
DEV void decode(uint32_t chunks, uint32_t &limb, uint32_t &i) { limb = blockIdx.x / chunks; i = (blockIdx.x % chunks) * blockDim.x + threadIdx.x; }

// op: 0 add, 1 sub, 2 negate(a).  a,b,out point at ciphertext arrays with `polys` polys each.
__global__ void k_addsub(const uint64_t *a, const uint64_t *b, uint64_t *out, const DevConsts *__restrict__ C, uint32_t chunks, int op) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint64_t q = C->q[limb % C->k].q; size_t o = (size_t)limb * C->n + i;
    uint64_t x = a[o];
    out[o] = op == 0 ? addmod(x, b[o], q) : (op == 1 ? submod(x, b[o], q) : negmod(x, q));
}
// out = sum_i in[idx[i]] ; limbs = polys*k of ONE ciphertext
__global__ void k_add_many(const uint64_t *in, const uint32_t *__restrict__ idx, uint32_t n_idx, size_t ct_words, uint64_t *out,
                           const DevConsts *__restrict__ C, uint32_t chunks) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint64_t q = C->q[limb % C->k].q; size_t o = (size_t)limb * C->n + i;
    uint64_t acc = 0;
    for (uint32_t t = 0; t < n_idx; t++) acc = addmod(acc, in[(size_t)idx[t] * ct_words + o], q);
    out[o] = acc;
}
DEV uint64_t scale_plain(const DevConsts *C, uint64_t m, uint32_t j) {       // Delta*m (+ r_t(q) in the upper half) mod q_j
    u128 p = (u128)C->delta[j] * m;
    if (m >= C->t_half) p += C->rtq[j];
    return bred128(p, C->q[j]);
}
// a: [count][polys][k][N]; pt: [..][N]; grid over count*polys*k limbs
__global__ void k_add_plain(const uint64_t *a, const uint64_t *pt, uint32_t pt_stride_words, uint64_t *out, const DevConsts *__restrict__ C,
                            uint32_t chunks, uint32_t polys, int subtract) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t k = C->k, j = limb % k, p = (limb / k) % polys, ct = limb / (k * polys);
    size_t o = (size_t)limb * C->n + i;
    uint64_t x = a[o];
    if (p == 0) {
        uint64_t s = scale_plain(C, pt[(size_t)ct * pt_stride_words + i], j), q = C->q[j].q;
        x = subtract ? submod(x, s, q) : addmod(x, s, q);
    }
    out[o] = x;
}
// lifted[pi][j][i] = fast plain lift of pt[pi][i] into q_j (multiply_plain)
__global__ void k_lift_plain(const uint64_t *pt, uint64_t *lifted, const DevConsts *__restrict__ C, uint32_t chunks, uint32_t pitch) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t k = C->k, j = limb % k, pi = limb / k;
    uint64_t m = pt[(size_t)pi * pitch * C->n + i];          // plaintext pi of the batch sits `pitch` plaintexts after plaintext pi-1
    lifted[(size_t)limb * C->n + i] = m >= C->t_half ? m + C->lift_inc[j] : m;
}
// x[ct][p][j][i] *= ptn[(ct*pstride)][j][i]   (both in NTT form)
__global__ void k_dyadic_pt(uint64_t *x, const uint64_t *ptn, uint32_t pstride, const DevConsts *__restrict__ C, uint32_t chunks, uint32_t polys) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t k = C->k, j = limb % k, ct = limb / (k * polys);
    size_t o = (size_t)limb * C->n + i;
    x[o] = mulmod(x[o], ptn[((size_t)ct * pstride * k + j) * C->n + i], C->q[j]);
}
// out[ct] = a[ct] * lifted scalar sc[ct*sstride*k + j]   (constant-plaintext multiply_plain)
__global__ void k_mul_scalar(const uint64_t *a, const uint64_t *__restrict__ sc, uint32_t sstride, uint64_t *out, const DevConsts *__restrict__ C,
                             uint32_t chunks, uint32_t polys) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t k = C->k, j = limb % k, ct = limb / (k * polys);
    size_t o = (size_t)limb * C->n + i;
    out[o] = mulmod(a[o], sc[(size_t)ct * sstride * k + j], C->q[j]);
}

// element-wise exact-FP64 modular arithmetic (per-coefficient kernels: GEMM fold, BEHZ extend / floor)
typedef ArF64T<1> BzF;
DEV double bz_canon(double x, const BzF::Mod &m) { double r = BzF::center(x, m); return r < 0.0 ? __dadd_rn(r, m.q) : r; }
// ------------------------------------------------------------------ HOT LOOP A: scalar GEMM
// Workgroup -> (coefficient chunk, limb, output tile mt, group g).  The mtiles output tiles of one (chunk, limb, g) read the SAME
// input elements; workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so the tiles take ids b, b+8, b+16, ...
// - same XCD, adjacent in time: the K input elements are fetched from HBM once and re-read from that XCD's L2 (dense 845->100:
// 5 tiles; without this every tile streamed the 528 MB of input ciphertexts again).
DEV void gemm_block_coords(uint32_t b, uint32_t chunks, uint32_t limbs, uint32_t mtiles, uint32_t G, uint32_t &chunk, uint32_t &limb, uint32_t &mt, uint32_t &g) {
    const uint32_t D = chunks * limbs * G;
    uint32_t d;
    if ((D & 7) == 0) { const uint32_t r = b >> 3; mt = r % mtiles; d = (r / mtiles) * 8 + (b & 7); }
    else { d = b % D; mt = b / D; }
    chunk = d % chunks; limb = (d / chunks) % limbs; g = d / (chunks * limbs);
}
// Group g gathers K input ciphertexts idx[g][:] once and produces M outputs (register tile MT):
//   out[out_idx[g][m]] = sum_k Wl[j][g][m][k] * in[idx[g][k]]  (+ scaled bias)   per limb j, coefficient i.
// Products accumulate lazily in 128 bits; one Barrett reduction per `lazy` terms.
template <int MT>
__global__ void __launch_bounds__(256) k_scalar_gemm(const uint64_t *__restrict__ in, const int32_t *__restrict__ idx, const uint64_t *__restrict__ Wl,
                                                     const int32_t *__restrict__ out_idx, const uint64_t *__restrict__ bias, const int32_t *__restrict__ bias_idx,
                                                     uint64_t *__restrict__ out, const DevConsts *__restrict__ C, uint32_t chunks, uint32_t G, uint32_t M,
                                                     uint32_t K, uint32_t mtiles, uint32_t lazy, uint32_t Kp, uint32_t obase) {
    const uint32_t n = C->n, k = C->k, limbs = 2 * k;
    uint32_t chunk, limb, mt, g;
    gemm_block_coords(blockIdx.x, chunks, limbs, mtiles, G, chunk, limb, mt, g);
    const uint32_t j = limb % k, i = chunk * blockDim.x + threadIdx.x;
    const size_t ctw = (size_t)limbs * n, e = (size_t)limb * n + i;
    const DMod qm = C->q[j];
    u128 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = 0;
    const int32_t *gi = idx + (size_t)g * Kp;                                            // row pitch Kp: 16 B aligned, -1 beyond K
    const uint64_t *gw = Wl + (((size_t)j * G + g) * mtiles + mt) * (size_t)K * MT;      // [kk][m]: the MT weights of a term are contiguous
    const uint32_t mcnt = min((uint32_t)MT, M - mt * MT);
    for (uint32_t k0 = 0; k0 < K; k0 += lazy) {            // reduction between blocks of `lazy` terms (see k_scalar_gemm_f64)
        const uint32_t k1 = min(K, k0 + lazy);
        for (uint32_t kk = k0; kk < k1; kk++) {
            const int32_t id = gi[kk];
            if (id < 0) continue;
            const uint64_t x = in[(size_t)id * ctw + e];
#pragma unroll
            for (int m = 0; m < MT; m++) acc[m] += (u128)x * gw[(size_t)kk * MT + m];       // zero-padded beyond mcnt
        }
        if (k1 < K) {
#pragma unroll
            for (int m = 0; m < MT; m++) acc[m] = bred128(acc[m], qm);
        }
    }
#pragma unroll
    for (int m = 0; m < MT; m++) {
        if ((uint32_t)m < mcnt && out_idx[g * M + mt * MT + m] >= 0) {          // -1: padding member of a smaller group
            const uint32_t o = g * M + mt * MT + m;
            uint64_t r = bred128(acc[m], qm);
            if (bias && limb < k) r = addmod(r, scale_plain(C, bias[(size_t)bias_idx[o] * n + i], j), qm.q);
            out[(size_t)(obase + (uint32_t)out_idx[o]) * ctw + e] = r;
        }
    }
}

// FP64 variant of the scalar GEMM for SMALL SIGNED weights (|w| < 2^20 after centring mod t - every PoolLayer weight
// round(w*scale) is): the input residue x < 2^(NL*LW) is split into NL limbs of LW bits, each limb times the weight is an
// exact double (< 2^(LW+20)), and up to 2^(52-LW-20) such terms accumulate exactly in one v_fma_f64 per limb - two half-rate
// FMAs per MAC instead of a 64x64->128-bit integer multiply-add (~12 half-rate instructions).  The signed weight is the same
// for every limb j, so the table is k times smaller as well.  Folded back with one 128-bit Barrett reduction per `lazy` terms.
// Weights through the vector memory path (MT = 20): a wave-uniform weight as a scalar operand means s_load, and scalar loads return
// out of order - the only possible wait is lgkmcnt(0), and ~70 free SGPRs hold less than two terms of 20 weights, so every other term
// exposed an L2 round trip (the 135 KB weight tile of a block never fits the 16 KB scalar cache).  Instead lanes 0..15 of every
// row of 16 load 16 consecutive table entries (5 coalesced 8 B loads per 4 terms, in-order vmcnt, requested two steps ahead) and
// the FMA takes its weight through DPP: v_fmac_f64_dpp ... row_newbcast:i reads src0 from lane i of the own row - the one DPP
// control gfx90a+ allows on FP64 instructions, at no extra issue slot.  (Through __builtin_amdgcn_update_dpp the compiler emits a
// separate v_mov_b64_dpp per weight, +50 % FP64-rate instructions - hence inline assembly; "s_nop 1" covers the 2 wait states a
// DPP read needs after a VALU write of its source, in case the register allocator put a copy right in front.)
#ifndef GEMM_DPP_W
#define GEMM_DPP_W 1
#endif
template <int LANE> DEV void fmac_bcast(double &acc, double w, double x) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x), "n"(LANE));
}
DEV void fmac_bcast_lane(int lane, double &acc, double w, double x) {          // `lane` is a constant after unrolling
    switch (lane) {
    case 0: fmac_bcast<0>(acc, w, x); break;   case 1: fmac_bcast<1>(acc, w, x); break;   case 2: fmac_bcast<2>(acc, w, x); break;   case 3: fmac_bcast<3>(acc, w, x); break;
    case 4: fmac_bcast<4>(acc, w, x); break;   case 5: fmac_bcast<5>(acc, w, x); break;   case 6: fmac_bcast<6>(acc, w, x); break;   case 7: fmac_bcast<7>(acc, w, x); break;
    case 8: fmac_bcast<8>(acc, w, x); break;   case 9: fmac_bcast<9>(acc, w, x); break;   case 10: fmac_bcast<10>(acc, w, x); break; case 11: fmac_bcast<11>(acc, w, x); break;
    case 12: fmac_bcast<12>(acc, w, x); break; case 13: fmac_bcast<13>(acc, w, x); break; case 14: fmac_bcast<14>(acc, w, x); break; default: fmac_bcast<15>(acc, w, x); break;
    }
}
// (178 VGPRs = 2 waves per SIMD.  Forcing 168 VGPRs for 3 waves costs 4 spilled registers and measured 16.3 vs 15.3 ms per batch.)
template <int MT, int NL, int LW>
__global__ void __launch_bounds__(256) k_scalar_gemm_f64(const uint64_t *__restrict__ in, const int32_t *__restrict__ idx, const double *__restrict__ Wd,
                                                         const int32_t *__restrict__ out_idx, const uint64_t *__restrict__ bias, const int32_t *__restrict__ bias_idx,
                                                         uint64_t *__restrict__ out, const DevConsts *__restrict__ C, uint32_t chunks, uint32_t G, uint32_t M,
                                                         uint32_t K, uint32_t mtiles, uint32_t lazy, uint32_t Kp, uint32_t obase) {
    const uint32_t n = C->n, k = C->k, limbs = 2 * k;
    uint32_t chunk, limb, mt, g;
    gemm_block_coords(blockIdx.x, chunks, limbs, mtiles, G, chunk, limb, mt, g);
    const uint32_t j = limb % k, i = chunk * blockDim.x + threadIdx.x;
    const size_t ctw = (size_t)limbs * n, e = (size_t)limb * n + i;
    const DMod qm = C->q[j];
    double acc[NL][MT], res[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        res[m] = 0.0;
#pragma unroll
        for (int l = 0; l < NL; l++) acc[l][m] = 0.0;
    }
    const int32_t *gi = idx + (size_t)g * Kp;                                            // row pitch Kp: 16 B aligned, -1 beyond K
    const double *gw = Wd + ((size_t)g * mtiles + mt) * (size_t)K * MT;                  // [kk][m], zero-padded
    const uint32_t mcnt = min((uint32_t)MT, M - mt * MT);
    // fold: value = sum_l acc[l] * 2^(l*LW) mod q_j, in exact FP64 (q_j < 2^49): Horner with one modular multiply per limb.  (The
    // 128-bit integer version - double -> int128 conversions and a Barrett reduction per output - cost more instructions than the 25
    // terms of a convolution window.)
    const BzF::Mod mq = {C->qd[j], C->qinvd[j]};
    auto fold = [&]() {
#pragma unroll
        for (int m = 0; m < MT; m++) {
            double r = acc[NL - 1][m];
            acc[NL - 1][m] = 0.0;
#pragma unroll
            for (int l = NL - 2; l >= 0; l--) { r = __dadd_rn(BzF::mulmod(r, (double)(1u << LW), mq), acc[l][m]); acc[l][m] = 0.0; }
            res[m] = BzF::center(__dadd_rn(res[m], r), mq);       // |res| <= q/2 between folds, |r| < 2^53 - q
        }
    };
    // blocks of `lazy` terms with the fold BETWEEN the inner loops: a fold test inside the term loop gets if-converted by the
    // compiler (the whole 128-bit fold executed every iteration under v_cndmask - 30x the instructions of the 2*MT FMAs)
    // Software pipeline: a term is one dependent scalar load (gather index) + one global load, ~1 us of latency against
    // 2*MT*NL FMAs.  Two register sets ping-pong: the input elements of the NEXT PF terms are requested before the FMAs of the
    // current PF terms are issued.  Fetch and compute are branch-free (padded taps and terms past the block read as x = 0 and
    // multiply whatever weight row follows - the table carries 8 spare rows), so the waits stay exact.
    constexpr int PF = 4;
    auto fetch = [&](uint64_t (&x)[PF], uint32_t kk, uint32_t k1) {
        // ONE 16 B scalar load for the PF gather indices (four dependent s_load_dword + wait chains cost more than the FMAs)
        const int4 ids = *reinterpret_cast<const int4 *>(__builtin_assume_aligned(gi + min(kk, Kp - 4), 16));
        const int32_t id[PF] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
        for (int p = 0; p < PF; p++) {
            const bool ok = kk + p < k1 && id[p] >= 0;
            const uint64_t v = in[(size_t)max(id[p], 0) * ctw + e];
            x[p] = ok ? v : 0;
        }
    };
    auto terms = [&](const uint64_t (&x)[PF], uint32_t kk) {
#pragma unroll
        for (int p = 0; p < PF; p++) {
            double xl[NL];
#pragma unroll
            for (int l = 0; l < NL; l++) xl[l] = (double)(uint32_t)((x[p] >> (l * LW)) & ((1ull << LW) - 1));
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const double w = gw[(size_t)(kk + p) * MT + m];
#pragma unroll
                for (int l = 0; l < NL; l++) acc[l][m] = __fma_rn(xl[l], w, acc[l][m]);
            }
        }
    };
    constexpr bool DPPW = GEMM_DPP_W && MT == 20 && PF == 4;
    constexpr int WV = DPPW ? PF * MT / 16 : 1;                    // 4 terms x 20 weights = 5 registers of 16 lanes
    auto wfetch = [&](double (&wv)[WV], uint32_t kk) {             // rows min(kk, K + 4) .. + 3: inside the 8 spare (zero) rows of the table
        const double *wp = gw + (size_t)min(kk, K + 4) * MT + (threadIdx.x & 15);
#pragma unroll
        for (int v = 0; v < WV; v++) wv[v] = wp[16 * v];
    };
    auto terms_dpp = [&](const uint64_t (&x)[PF], double (&wv)[WV]) {
#pragma unroll
        for (int p = 0; p < PF; p++) {
            double xl[NL];
#pragma unroll
            for (int l = 0; l < NL; l++) xl[l] = (double)(uint32_t)((x[p] >> (l * LW)) & ((1ull << LW) - 1));
            asm volatile("s_nop 1");
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const int f = p * MT + m;
#pragma unroll
                for (int l = 0; l < NL; l++) fmac_bcast_lane(f & 15, acc[l][m], wv[(f >> 4) % WV], xl[l]);
            }
        }
    };
    for (uint32_t k0 = 0; k0 < K; k0 += lazy) {
        const uint32_t k1 = min(K, k0 + lazy);
        uint64_t xa[PF], xb[PF];
        if constexpr (DPPW) {
            double wa[WV], wb[WV];
            fetch(xa, k0, k1); wfetch(wa, k0);
            for (uint32_t kk = k0; kk < k1; kk += 2 * PF) {
                fetch(xb, kk + PF, k1); wfetch(wb, kk + PF);
                terms_dpp(xa, wa);
                fetch(xa, kk + 2 * PF, k1); wfetch(wa, kk + 2 * PF);
                terms_dpp(xb, wb);
            }
        } else {
            fetch(xa, k0, k1);
            for (uint32_t kk = k0; kk < k1; kk += 2 * PF) {
                fetch(xb, kk + PF, k1);
                terms(xa, kk);
                fetch(xa, kk + 2 * PF, k1);
                terms(xb, kk + PF);
            }
        }
        fold();
    }
#pragma unroll
    for (int m = 0; m < MT; m++) {
        if ((uint32_t)m < mcnt && out_idx[g * M + mt * MT + m] >= 0) {
            const uint32_t o = g * M + mt * MT + m;
            uint64_t r = BzF::to_u64(res[m], mq);
            if (bias && limb < k) r = addmod(r, scale_plain(C, bias[(size_t)bias_idx[o] * n + i], j), qm.q);
            out[(size_t)(obase + (uint32_t)out_idx[o]) * ctw + e] = r;
        }
    }
}

// ------------------------------------------------------------------ HOT LOOP B: BEHZ multiply
// Step 0/1 (fastbconv_mtilde + mont_rq): q -> Bsk, removing q-overflows via m~ = 2^32.
// src ciphertext c = srcbase + (first + c*stride)*ctw ; writes the q copy (for the q-side NTT) and Bsk.
template <int K>
__global__ void __launch_bounds__(256) k_behz_extend(const uint64_t *__restrict__ src, uint32_t stride, uint64_t *__restrict__ aq, uint64_t *__restrict__ ab,
                                                     const DevConsts *__restrict__ C, uint32_t chunks) {
    const uint32_t n = C->n;
    const uint32_t cp = blockIdx.x / chunks, i = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;   // cp = ct*2 + poly
    const uint32_t ct = cp >> 1, p = cp & 1;
    const uint64_t *x = src + ((size_t)ct * stride * 2 + p) * K * n + i;
    uint64_t *oq = aq + (size_t)cp * K * n + i, *ob = ab + (size_t)cp * (K + 1) * n + i;
    uint64_t y[K], mt = 0;
#pragma unroll
    for (int j = 0; j < K; j++) {
        uint64_t v = x[(size_t)j * n];
        oq[(size_t)j * n] = v;
        y[j] = mulmod(v, C->mt_inv_qhat_q[j], C->q[j]);
        mt += y[j] * C->qhat_mt[j];
    }
    mt &= 0xffffffffull;
    const uint64_t r = (0 - mt * C->inv_q_mt) & 0xffffffffull;      // r = -x q^{-1} mod m~
#pragma unroll
    for (int b = 0; b <= K; b++) {
        const DMod bm = C->bsk[b];
        const uint64_t rr = r >= 0x80000000ull ? r + bm.q - 0x100000000ull : r;     // centred r as a residue mod b
        u128 acc = (u128)rr * C->ex_R_bsk[b];                                      // (x_b + q*r) * m~^-1 with the factors folded:
#pragma unroll
        for (int j = 0; j < K; j++) acc += (u128)y[j] * C->ex_Q_bsk[b][j];         // one lazy accumulation, one Barrett reduction
        ob[(size_t)b * n] = bred128(acc, bm);
    }
}
// The same two element-wise steps in exact FP64 (DevConsts::behz_f64: every data and auxiliary prime < 2^49).  The integer
// versions spend ~2700 / ~1450 VALU instructions per coefficient, mostly in 128-bit Barrett reductions and 64x64->128
// products assembled from 32-bit pieces; here a term of a base conversion is ArF64::mulmod (6 FP64 instructions, |r| <= 2.1 p), a
// conversion of <= 6 terms is summed exactly in one double and reduced once.  Representatives: the CRT coefficients
// y_j = [x (q/q_j)^-1]_{q_j} MUST be canonical (another representative changes the q-overflow count and with it the words
// SEAL produces); residues that are only re-reduced (f_b, z_j, alpha) may stay lazy - z_j + s b_j shifts the Shenoy-Kumaresan
// sum by s B and alpha by s, which cancels.
template <int K>
__global__ void __launch_bounds__(256) k_behz_extend_f64(const uint64_t *__restrict__ src, uint32_t stride, uint64_t *__restrict__ aq, uint64_t *__restrict__ ab,
                                                         const DevConsts *__restrict__ C, uint32_t chunks) {
    const uint32_t n = C->n;
    const uint32_t cp = blockIdx.x / chunks, i = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;   // cp = ct*2 + poly
    const uint32_t ct = cp >> 1, p = cp & 1;
    const uint64_t *x = src + ((size_t)ct * stride * 2 + p) * K * n + i;
    uint64_t *oq = aq + (size_t)cp * K * n + i, *ob = ab + (size_t)cp * (K + 1) * n + i;
    double y[K];
    uint32_t mt = 0;
#pragma unroll
    for (int j = 0; j < K; j++) {
        const uint64_t v = x[(size_t)j * n];
        if (aq) oq[(size_t)j * n] = v;             // aq == nullptr: the fused squaring kernel reads the q limbs from the ciphertext itself
        const BzF::Mod mq = {C->qd[j], C->qinvd[j]};
        y[j] = bz_canon(BzF::mulmod(BzF::from_u64(v), C->bd.mt_inv_qhat_q[j], mq), mq);       // canonical: feeds the mod-m~ sum
        mt += (uint32_t)__double_as_longlong(__dadd_rn(y[j], 4503599627370496.0)) * (uint32_t)C->qhat_mt[j];   // low 32 bits of y (exact integer < 2^49)
    }
    const double r = (double)(int32_t)(0u - mt * (uint32_t)C->inv_q_mt);                       // centred r = -x q^-1 mod m~ = 2^32
#pragma unroll
    for (int b = 0; b <= K; b++) {
        const BzF::Mod mb = {C->qd[K + b], C->qinvd[K + b]};
        double acc = BzF::mulmod(r, C->bd.ex_R_bsk[b], mb);
#pragma unroll
        for (int j = 0; j < K; j++) acc = __dadd_rn(acc, BzF::mulmod(y[j], C->bd.ex_Q_bsk[b][j], mb));
        ob[(size_t)b * n] = BzF::to_u64(acc, mb);
    }
}
template <int K>
__global__ void __launch_bounds__(256) k_behz_floor_f64(const uint64_t *__restrict__ dq, const uint64_t *__restrict__ db, uint64_t *__restrict__ out,
                                                        const DevConsts *__restrict__ C, uint32_t chunks) {
    const uint32_t n = C->n;
    const uint32_t cp = blockIdx.x / chunks, i = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;   // cp = ct*3 + poly
    const uint64_t *xq = dq + (size_t)cp * K * n + i, *xb = db + (size_t)cp * (K + 1) * n + i;
    uint64_t *o = out + (size_t)cp * K * n + i;
    double y[K], f[K + 1], z[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
        const BzF::Mod mq = {C->qd[j], C->qinvd[j]};
        y[j] = bz_canon(BzF::mulmod(BzF::from_u64(xq[(size_t)j * n]), C->bd.fl_c1_q[j], mq), mq);     // [x t (q/q_j)^-1]_{q_j}, canonical
    }
#pragma unroll
    for (int b = 0; b <= K; b++) {
        const BzF::Mod mb = {C->qd[K + b], C->qinvd[K + b]};
        double acc = BzF::mulmod(BzF::from_u64(xb[(size_t)b * n]), C->bd.fl_T_bsk[b], mb);          // (x_b t - conv_b) q^-1, folded
#pragma unroll
        for (int j = 0; j < K; j++) acc = __dadd_rn(acc, BzF::mulmod(y[j], C->bd.fl_N_bsk[b][j], mb));
        f[b] = acc;                                                                                    // |f| <= 12.6 b < 2^53
    }
    const BzF::Mod msk = {C->qd[2 * K], C->qinvd[2 * K]};
    double acc = BzF::mulmod(-f[K], C->bd.inv_B_msk, msk);
#pragma unroll
    for (int j = 0; j < K; j++) {
        const BzF::Mod mb = {C->qd[K + j], C->qinvd[K + j]};
        z[j] = BzF::mulmod(f[j], C->bd.inv_bhat_b[j], mb);
        acc = __dadd_rn(acc, BzF::mulmod(z[j], C->bd.fl_A_msk[j], msk));
    }
    const double alpha = BzF::center(acc, msk);                                                        // centred alpha_sk
#pragma unroll
    for (int j = 0; j < K; j++) {
        const BzF::Mod mq = {C->qd[j], C->qinvd[j]};
        double a2 = BzF::mulmod(-alpha, C->bd.B_q[j], mq);
#pragma unroll
        for (int l = 0; l < K; l++) a2 = __dadd_rn(a2, BzF::mulmod(z[l], C->bd.bhat_q[j][l], mq));
        o[(size_t)j * n] = BzF::to_u64(a2, mq);
    }
}
// Step 2: tensor product in NTT form; A,B: [cnt][2][L][N], D: [cnt][3][L][N]; L limbs with moduli base_off..
__global__ void k_tensor(const uint64_t *__restrict__ A, const uint64_t *__restrict__ B, uint64_t *__restrict__ D, const DevConsts *__restrict__ C,
                         uint32_t chunks, uint32_t L, uint32_t base_off) {
    uint32_t limb, i; decode(chunks, limb, i);                       // limb = ct*L + l
    const uint32_t n = C->n, l = limb % L, ct = limb / L, mod = base_off + l;
    const DMod m = mod < C->k ? C->q[mod] : C->bsk[mod - C->k];
    const size_t Ln = (size_t)L * n, a = (size_t)ct * 2 * Ln + (size_t)l * n + i, d = (size_t)ct * 3 * Ln + (size_t)l * n + i;
    uint64_t a0 = A[a], a1 = A[a + Ln], b0 = B[a], b1 = B[a + Ln];
    D[d] = mulmod(a0, b0, m);
    D[d + Ln] = addmod(mulmod(a0, b1, m), mulmod(a1, b0, m), m.q);
    D[d + 2 * Ln] = mulmod(a1, b1, m);
}
// Steps 3/4 (x t, fast_floor: q u Bsk -> Bsk, fastbconv_sk: Bsk -> q) per coefficient.
template <int K>
__global__ void __launch_bounds__(256) k_behz_floor(const uint64_t *__restrict__ dq, const uint64_t *__restrict__ db, uint64_t *__restrict__ out,
                                                    const DevConsts *__restrict__ C, uint32_t chunks) {
    const uint32_t n = C->n;
    const uint32_t cp = blockIdx.x / chunks, i = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;   // cp = ct*3 + poly
    const uint64_t *xq = dq + (size_t)cp * K * n + i, *xb = db + (size_t)cp * (K + 1) * n + i;
    uint64_t *o = out + (size_t)cp * K * n + i;
    uint64_t y[K], f[K + 1], z[K];
#pragma unroll
    for (int j = 0; j < K; j++) y[j] = mulmod(xq[(size_t)j * n], C->fl_c1_q[j], C->q[j]);           // x * t * (q/q_j)^-1
#pragma unroll
    for (int b = 0; b <= K; b++) {
        const DMod bm = C->bsk[b];
        u128 acc = (u128)xb[(size_t)b * n] * C->fl_T_bsk[b];                                       // (x_b*t - conv_b) * q^-1, folded
#pragma unroll
        for (int j = 0; j < K; j++) acc += (u128)y[j] * C->fl_N_bsk[b][j];
        f[b] = bred128(acc, bm);
    }
    const DMod sk = C->bsk[K];
    u128 acc = (u128)(sk.q - f[K]) * C->inv_B_msk;
#pragma unroll
    for (int j = 0; j < K; j++) { z[j] = mulmod(f[j], C->inv_bhat_b[j], C->bsk[j]); acc += (u128)z[j] * C->fl_A_msk[j]; }
    const uint64_t alpha = bred128(acc, sk);
    const bool neg = alpha > (sk.q >> 1);
#pragma unroll
    for (int j = 0; j < K; j++) {
        const DMod qm = C->q[j];
        u128 a2 = neg ? (u128)C->B_q[j] * (sk.q - alpha) : (u128)(qm.q - C->B_q[j]) * alpha;        // -alpha*B (centred alpha)
#pragma unroll
        for (int l = 0; l < K; l++) a2 += (u128)z[l] * C->bhat_q[j][l];
        o[(size_t)j * n] = bred128(a2, qm);
    }
}

// ------------------------------------------------------------------ key switching (relinearize / Galois)
// Fused per (ciphertext, output limb j): for every (source limb l, digit d): extract the base-2^dbc digit of
// target[l], NTT it under q_j in LDS, multiply-accumulate with the key pair K[(l,d)][0/1][j] (NTT form) in
// registers; finally INTT both accumulators and add them to add0/add1.  The digit polynomials never touch HBM.
// block = NT threads, EPT = N/NT accumulators per thread per output poly.
template <int EPT>
__global__ void __launch_bounds__(1024) k_keyswitch(const uint64_t *__restrict__ target, size_t tgt_stride, const uint64_t *__restrict__ add0,
                                                    const uint64_t *__restrict__ add1, size_t add_stride, const uint64_t *__restrict__ key,
                                                    uint64_t *__restrict__ out, const DevConsts *__restrict__ C, int galois) {
    extern __shared__ uint64_t s[];
    const uint32_t n = C->n, k = C->k, nt = blockDim.x, tid = threadIdx.x;
    const uint32_t ct = blockIdx.x / k, j = blockIdx.x % k;
    const DMod qm = C->q[j];
    const uint64_t q = qm.q;
    const uint64_t *tw = tw_of(C, j);
    const int dbc = galois ? C->gdbc : C->dbc;
    const uint64_t mask = (1ull << dbc) - 1;
    const size_t kn = (size_t)k * n;
    uint64_t acc0[EPT], acc1[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) { acc0[e] = 0; acc1[e] = 0; }
    const uint64_t *kp = key;
    for (uint32_t l = 0; l < k; l++) {
        const uint32_t nd = galois ? C->gk_dig[l] : C->rl_dig[l];
        const uint64_t *src = target + (size_t)ct * tgt_stride + (size_t)l * n;
        for (uint32_t d = 0; d < nd; d++, kp += 2 * kn) {
            const int sh = dbc * (int)d;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                uint64_t v = (src[tid + e * nt] >> sh) & mask;
                if (mask >= q) v = v >= q ? bred128(v, 0, qm) : v;
                s[tid + e * nt] = v;
            }
            __syncthreads();
            ntt_fwd_lds(s, tw, tw + n, q, n);
            const uint64_t *k0 = kp + (size_t)j * n, *k1 = kp + kn + (size_t)j * n;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                uint64_t x = canon4(s[tid + e * nt], q);
                acc0[e] = addmod(acc0[e], mulmod(x, k0[tid + e * nt], qm), q);
                acc1[e] = addmod(acc1[e], mulmod(x, k1[tid + e * nt], qm), q);
            }
            __syncthreads();
        }
    }
    const uint64_t ni = C->ninv[j], nis = C->ninvs[j];
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int e = 0; e < EPT; e++) s[tid + e * nt] = p ? acc1[e] : acc0[e];
        __syncthreads();
        ntt_inv_lds(s, tw + 2 * (size_t)n, tw + 3 * (size_t)n, q, n);
        const uint64_t *ad = p ? add1 : add0;
        uint64_t *o = out + ((size_t)ct * 2 + p) * kn + (size_t)j * n;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            uint64_t v = shoup_lazy(s[tid + e * nt], ni, nis, q);
            v = v >= q ? v - q : v;
            if (ad) v = addmod(v, ad[(size_t)ct * add_stride + (size_t)j * n + tid + e * nt], q);
            o[tid + e * nt] = v;
        }
        __syncthreads();
    }
}
// Galois automorphism x -> x^elt on coefficient-form limbs: dst[(i*elt) mod N] = +-src[i]
__global__ void k_galois(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, const DevConsts *__restrict__ C, uint32_t chunks, uint64_t elt) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t n = C->n;
    const uint64_t q = C->q[limb % C->k].q;
    const uint64_t raw = (uint64_t)i * elt;
    const uint32_t idx = (uint32_t)(raw & (n - 1));
    uint64_t v = src[(size_t)limb * n + i];
    dst[(size_t)limb * n + idx] = ((raw >> C->logn) & 1) ? negmod(v, q) : v;
}

// The same permutation with both global accesses coalesced: one workgroup per limb stages it in LDS - coalesced 8 B/lane loads, LDS
// writes at the permuted positions (odd stride -> bank-conflict free), barrier, linear LDS reads, coalesced stores.  The scattered
// global stores of k_galois reach 1.6 TB/s (a 64 B sector per lane and instruction); a batched rotation at N = 16384 spent 21 % there.
__global__ void __launch_bounds__(1024) k_galois_lds(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, const DevConsts *__restrict__ C, uint64_t elt) {
    extern __shared__ uint64_t gs[];
    const uint32_t n = C->n, limb = blockIdx.x, logn = C->logn;
    const uint64_t q = C->q[limb % C->k].q;
    const uint64_t *x = src + (size_t)limb * n;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t raw = (uint64_t)i * elt, v = x[i];
        gs[(uint32_t)(raw & (n - 1))] = ((raw >> logn) & 1) ? negmod(v, q) : v;
    }
    __syncthreads();
    uint64_t *o = dst + (size_t)limb * n;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) o[i] = gs[i];
}

// ------------------------------------------------------------------ register-radix NTT kernels (N = 2^L, L = 10..14)
DEV uint64_t modulus_of(const DevConsts *C, uint32_t mod) { return mod < C->k ? C->q[mod].q : (mod < C->k + C->kb ? C->bsk[mod - C->k].q : C->t.q); }

// per-policy views of the context constants
template <class AR> struct ArCtx;
template <> struct ArCtx<ArU64> {
    ArU64::Mod m; ArU64::Tw fw, iv; uint64_t ni, nis;
    DEV ArCtx(const DevConsts *C, uint32_t mod) {
        const uint64_t q = modulus_of(C, mod); const uint64_t *tw = tw_of(C, mod); const size_t n = C->n;
        typedef const NTT_GLOBAL uint64_t *GP;
        m = {q, 2 * q}; fw = {(GP)tw, (GP)(tw + n)}; iv = {(GP)(tw + 2 * n), (GP)(tw + 3 * n)}; ni = C->ninv[mod]; nis = C->ninvs[mod];
    }
    DEV uint64_t load(uint64_t v) const { return v; }
    DEV uint64_t canon(uint64_t v) const { return canon4(v, m.q); }                       // forward output in [0,4q)
    DEV uint64_t scaled(uint64_t v) const { uint64_t o = shoup_lazy(v, ni, nis, m.q); return o >= m.q ? o - m.q : o; }   // * N^-1, canonical
};
template <int RN> struct ArCtx<ArF64T<RN>> {
    typedef ArF64T<RN> ArF64;
    typename ArF64::Mod m; typename ArF64::Tw fw, iv; double ni;
    DEV ArCtx(const DevConsts *C, uint32_t mod) {
        const double *tw = C->twd + (size_t)mod * 2 * C->n;
        typedef const NTT_GLOBAL double *GP;
        m = {C->qd[mod], C->qinvd[mod]}; fw = {(GP)tw}; iv = {(GP)(tw + C->n)}; ni = C->ninvd[mod];
    }
    DEV double load(uint64_t v) const { return ArF64::from_u64(v); }
    DEV uint64_t canon(double v) const { return ArF64::to_u64(v, m); }
    DEV uint64_t scaled(double v) const { return ArF64::to_u64(ArF64::mulmod(v, ni, m), m); }
};

template <int L, class AR, bool inverse>
__global__ void __launch_bounds__(NttPlan<L>::NT) k_ntt_rr(uint64_t *data, const DevConsts *__restrict__ C, uint32_t base_off, uint32_t nmod) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t tid = threadIdx.x, mod = base_off + blockIdx.x % nmod;
    const ArCtx<AR> A(C, mod);
    uint64_t *x = data + (size_t)blockIdx.x * n;
    T v[16];
    if (!inverse) {
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = A.load(x[pass_index<L, SA, 0>(tid, r)]);
        ntt_forward_regs<AR, L>(v, s, A.fw, A.m, tid);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            ulonglong2 o; o.x = A.canon(v[r]); o.y = A.canon(v[r + 1]);
            *reinterpret_cast<ulonglong2 *>(x + tail_index<L>(tid, r)) = o;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            ulonglong2 i2 = *reinterpret_cast<const ulonglong2 *>(x + tail_index<L>(tid, r));
            v[r] = A.load(i2.x); v[r + 1] = A.load(i2.y);
        }
        ntt_inverse_regs<AR, L>(v, s, A.iv, A.m, tid);
#pragma unroll
        for (int r = 0; r < 16; r++) x[pass_index<L, SA, 0>(tid, r)] = A.scaled(v[r]);
    }
}

// BEHZ step 2 fused into the inverse transform: block = (ciphertext, output poly p of the tensor product, limb).  The NTT-form
// operands are read at the positions the inverse transform starts from (16 B/lane), d_p = a0*b0 | a0*b1 + a1*b0 | a1*b1 is formed
// in registers and transformed back at once - the 3-poly NTT-form tensor never exists in HBM (saves one write + one read of
// 3(k + k+1) limbs per ciphertext and a kernel).  A, B: [cnt][2][Lm][N]; D: [cnt][3][Lm][N] (coefficient form, canonical).
template <class AR> struct TensorOps;
template <> struct TensorOps<ArU64> {
    DMod dm;
    DEV TensorOps(const DevConsts *C, uint32_t mod) { dm = mod < C->k ? C->q[mod] : C->bsk[mod - C->k]; }
    DEV uint64_t mul(uint64_t a, uint64_t b, const ArCtx<ArU64> &) const { return mulmod(a, b, dm); }
    DEV uint64_t add(uint64_t a, uint64_t b) const { return addmod(a, b, dm.q); }
};
template <int RN> struct TensorOps<ArF64T<RN>> {
    DEV TensorOps(const DevConsts *, uint32_t) {}
    DEV double mul(double a, double b, const ArCtx<ArF64T<RN>> &A) const { return ArF64T<RN>::mulmod(a, b, A.m); }
    DEV double add(double a, double b) const { return __dadd_rn(a, b); }
};
// Dense MultiplyPlain on the register-radix core, two launches instead of six (lift, transform, copy, transform, dyadic, transform):
//   k_lift_ntt:        block = (plaintext, limb j): coefficients mod t -> fast plain lift into q_j -> forward transform -> NTT form
//   k_mul_plain_fused: block = (ciphertext, poly, limb j): forward transform, pointwise product with the plaintext's NTT form (read at
//                      the positions the thread holds, 16 B/lane), inverse transform, N^-1, store.  The product never exists in HBM in
//                      NTT form: read ct limb + plaintext limb, write ct limb (3 limb transfers instead of 9).  src_stride = 0
//                      broadcasts ONE input ciphertext over all plaintexts (the row-dot batches of the LoLa dense layers: 5488 rows at
//                      CIFAR shapes, previously 5488 device-to-device copies per call).
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT) k_lift_ntt(const uint64_t *__restrict__ pt, uint32_t pitch, uint64_t *__restrict__ lifted, const DevConsts *__restrict__ C) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t tid = threadIdx.x, k = C->k, j = blockIdx.x % k, pi = blockIdx.x / k;
    const ArCtx<AR> A(C, j);
    const uint64_t *x = pt + (size_t)pi * pitch * n, th = C->t_half, inc = C->lift_inc[j];
    T v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { const uint64_t m = x[pass_index<L, SA, 0>(tid, r)]; v[r] = A.load(m >= th ? m + inc : m); }
    ntt_forward_regs<AR, L>(v, s, A.fw, A.m, tid);
    uint64_t *o = lifted + (size_t)blockIdx.x * n;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        ulonglong2 w; w.x = A.canon(v[r]); w.y = A.canon(v[r + 1]);
        *reinterpret_cast<ulonglong2 *>(o + tail_index<L>(tid, r)) = w;
    }
}
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT) k_mul_plain_fused(const uint64_t *src, size_t src_stride, const uint64_t *__restrict__ ptn, uint32_t pstride,
                                                                    uint64_t *out, const DevConsts *__restrict__ C, uint32_t polys) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t tid = threadIdx.x, k = C->k, j = blockIdx.x % k, cp = blockIdx.x / k, ct = cp / polys, p = cp % polys;
    const ArCtx<AR> A(C, j);
    const TensorOps<AR> ops(C, j);
    const uint64_t *x = src + (size_t)ct * src_stride + ((size_t)p * k + j) * n;
    const uint64_t *w = ptn + ((size_t)ct * pstride * k + j) * n;
    T v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = A.load(x[pass_index<L, SA, 0>(tid, r)]);
    ntt_forward_regs<AR, L>(v, s, A.fw, A.m, tid);
    uint32_t tm = tid;
    asm volatile("" : "+v"(tm));
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const ulonglong2 y = *reinterpret_cast<const ulonglong2 *>(w + tail_index<L>(tm, r));
        if constexpr (std::is_same<T, double>::value) {                   // lazy transform output x canonical plaintext word: exact (see KsMac)
            v[r] = ops.mul(v[r], A.load(y.x), A); v[r + 1] = ops.mul(v[r + 1], A.load(y.y), A);
        } else {
            v[r] = ops.mul(A.canon(v[r]), y.x, A); v[r + 1] = ops.mul(A.canon(v[r + 1]), y.y, A);
        }
    }
    if (!ntt_tail_local<L>()) __syncthreads();                            // (block-local tail: the inverse starts inside the wave's own blocks)
    uint32_t ti = tid;
    asm volatile("" : "+v"(ti));
    ntt_inverse_regs<AR, L>(v, s, A.iv, A.m, ti);
    uint64_t *o = out + ((size_t)cp * k + j) * n;
#pragma unroll
    for (int r = 0; r < 16; r++) o[pass_index<L, SA, 0>(ti, r)] = A.scaled(v[r]);
}

template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT) k_intt_tensor(const uint64_t *__restrict__ A_, const uint64_t *__restrict__ B_, uint64_t *__restrict__ D,
                                                                const DevConsts *__restrict__ C, uint32_t base_off, uint32_t Lm) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t tid = threadIdx.x;
    const uint32_t l = blockIdx.x % Lm, p = (blockIdx.x / Lm) % 3, ct = blockIdx.x / (3 * Lm), mod = base_off + l;
    const ArCtx<AR> A(C, mod);
    const TensorOps<AR> ops(C, mod);
    const size_t Ln = (size_t)Lm * n;
    const uint64_t *a0 = A_ + (size_t)ct * 2 * Ln + (size_t)l * n, *a1 = a0 + Ln;
    const uint64_t *b0 = B_ + (size_t)ct * 2 * Ln + (size_t)l * n, *b1 = b0 + Ln;
    T v[16];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const uint32_t pos = tail_index<L>(tid, r);
        if (p == 0) {
            const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(a0 + pos), y = *reinterpret_cast<const ulonglong2 *>(b0 + pos);
            v[r] = ops.mul(A.load(x.x), A.load(y.x), A); v[r + 1] = ops.mul(A.load(x.y), A.load(y.y), A);
        } else if (p == 2) {
            const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(a1 + pos), y = *reinterpret_cast<const ulonglong2 *>(b1 + pos);
            v[r] = ops.mul(A.load(x.x), A.load(y.x), A); v[r + 1] = ops.mul(A.load(x.y), A.load(y.y), A);
        } else {
            const ulonglong2 x0 = *reinterpret_cast<const ulonglong2 *>(a0 + pos), y1 = *reinterpret_cast<const ulonglong2 *>(b1 + pos);
            const ulonglong2 x1 = *reinterpret_cast<const ulonglong2 *>(a1 + pos), y0 = *reinterpret_cast<const ulonglong2 *>(b0 + pos);
            v[r] = ops.add(ops.mul(A.load(x0.x), A.load(y1.x), A), ops.mul(A.load(x1.x), A.load(y0.x), A));
            v[r + 1] = ops.add(ops.mul(A.load(x0.y), A.load(y1.y), A), ops.mul(A.load(x1.y), A.load(y0.y), A));
        }
    }
    ntt_inverse_regs<AR, L>(v, s, A.iv, A.m, tid);
    uint64_t *o = D + ((size_t)ct * 3 + p) * Ln + (size_t)l * n;
#pragma unroll
    for (int r = 0; r < 16; r++) o[pass_index<L, SA, 0>(tid, r)] = A.scaled(v[r]);
}

// BEHZ steps 2-4 of a SQUARING in one kernel (FP64 policies): block = (ciphertext, limb) of the q or the Bsk base.  The operand polys
// a0, a1 arrive in COEFFICIENT form (k_behz_extend's output) and the tensor (a0^2, 2 a0 a1, a1^2) leaves in coefficient form:
//     step 0:  A0 = NTT(a0) -> parked in d1's place;               d0 = INTT(A0^2)
//     step 1:  A1 = NTT(a1) -> parked in d2's place;  A0 back;     d1 = INTT(2 A0 A1)   (overwrites the parked A0)
//     step 2:  A1 back;                                            d2 = INTT(A1^2)      (overwrites the parked A1)
// Two forward and three inverse transforms like the separate launches, but every step needs only the 16 coefficients of ONE polynomial
// per thread (126 VGPRs, two resident workgroups per CU) - keeping A0 and A1 in registers for the cross term would cost 64 more VGPRs
// and a workgroup per CU.  The parked values are the thread's own doubles at its own 16 B/lane positions (stored and re-read by the
// same thread: program order), 64 KiB per polynomial that is overwritten by the result a few microseconds later.  Measured HBM traffic
// (profiles/r01_pmc_square_fused.txt): the parked limbs ARE written back before they are overwritten (5 limbs written per block, not 3)
// and part of their re-reads comes from HBM (3.2-4.4 limbs read) - against 2R + 2W (forward transforms in place) + 4R + 3W (tensor +
// inverse transforms) of the separate launches.
// The inverse transform's workgroup barrier orders "everybody has re-read its parked words" before any result word is stored over them.
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT, 4) k_square_fused(const uint64_t *__restrict__ A_, size_t a_stride, uint64_t *__restrict__ D,
                                                                    const DevConsts *__restrict__ C, uint32_t base_off, uint32_t Lm) {
    // 4 waves per SIMD: two 512-thread workgroups per CU.  a_stride: words between the operands of consecutive ciphertexts (the q side
    // reads the input ciphertexts in place, the Bsk side k_behz_extend's array)
    typedef typename AR::T T;
    static_assert(std::is_same<T, double>::value, "FP64 policies only");
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t tid = threadIdx.x;
    const uint32_t l = blockIdx.x % Lm, ct = blockIdx.x / Lm, mod = base_off + l;
    const ArCtx<AR> A(C, mod);
    const size_t Ln = (size_t)Lm * n;
    const uint64_t *a0 = A_ + (size_t)ct * a_stride + (size_t)l * n, *a1 = a0 + Ln;
    uint64_t *d0 = D + (size_t)ct * 3 * Ln + (size_t)l * n, *d1 = d0 + Ln, *d2 = d1 + Ln;
#pragma unroll 1
    for (int step = 0; step < 3; step++) {
        uint32_t tl = tid;
        asm volatile("" : "+v"(tl));                             // one transform's address math / twiddles live at a time
        T v[16];
        if (step < 2) {
            const uint64_t *x = step ? a1 : a0;
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = A.load(x[pass_index<L, SA, 0>(tl, r)]);
            ntt_forward_regs<AR, L, true>(v, s, A.fw, A.m, tl);  // PRE: the image of the previous inverse transform is free
            AR::renorm(v, A.m);                                  // lazy transform output (up to 28 q) -> |x| <= q/2
            uint64_t *park = step ? d2 : d1;                     // bit patterns of the doubles: every access to D stays a u64 access
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                ulonglong2 w; w.x = (uint64_t)__double_as_longlong(v[r]); w.y = (uint64_t)__double_as_longlong(v[r + 1]);
                *reinterpret_cast<ulonglong2 *>(park + tail_index<L>(tl, r)) = w;
            }
            if (step == 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = AR::mulmod(v[r], v[r], A.m);
            } else {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(d1 + tail_index<L>(tl, r));
                    const T wa = __longlong_as_double((long long)w.x), wb = __longlong_as_double((long long)w.y);
                    v[r] = AR::mulmod(__dadd_rn(wa, wa), v[r], A.m); v[r + 1] = AR::mulmod(__dadd_rn(wb, wb), v[r + 1], A.m);
                }
            }
            if (!ntt_tail_local<L>()) __syncthreads();           // (block-local tail: the inverse starts inside the wave's own blocks)
        } else {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(d2 + tail_index<L>(tl, r));
                const T wa = __longlong_as_double((long long)w.x), wb = __longlong_as_double((long long)w.y);
                v[r] = AR::mulmod(wa, wa, A.m); v[r + 1] = AR::mulmod(wb, wb, A.m);
            }
            __syncthreads();                                     // no forward transform in this step: the previous inverse's image is free
        }
        uint32_t ti = tid;
        asm volatile("" : "+v"(ti));                             // the inverse transform's address math starts here, not before the forward one
        ntt_inverse_regs<AR, L>(v, s, A.iv, A.m, ti);
        uint64_t *o = step == 0 ? d0 : (step == 1 ? d1 : d2);
        uint32_t to = tid;
        asm volatile("" : "+v"(to));
#pragma unroll
        for (int r = 0; r < 16; r++) o[pass_index<L, SA, 0>(to, r)] = A.scaled(v[r]);
    }
}

// Key switching on the register-radix core: block = (ciphertext, output limb j).  For every (source limb l, digit d) the
// base-2^dbc digit of the 16 coefficients a thread owns goes through the forward transform in registers/LDS and is
// multiply-accumulated with the key pair (16 B/lane coalesced key loads) into per-thread accumulators; two inverse transforms
// finish and the result is added to (add0, add1).  The digit polynomials never exist in HBM.
// U64 policy: keys are u64 residues, canonical accumulators.  F64 policy: keys were converted to doubles at upload,
// accumulators are lazy doubles recentred every `accmax` terms.
// forward-transform policy of the key-switch kernels: FP64 policies read their twiddles from an LDS copy of the table
// (N <= 8192: image + table = 132 KiB of the 160 KiB; the N = 16384 image alone is 136 KiB)
template <class AR, int L> struct KsFwd { typedef AR P; static constexpr bool lds = false; };
template <int RN, int L> struct KsFwd<ArF64T<RN>, L> { typedef typename std::conditional<(L <= 13), ArF64LdsT<RN>, ArF64T<RN>>::type P; static constexpr bool lds = L <= 13; };
// copy `words` doubles of a global table behind the exchange image (all threads; caller synchronises)
DEV void stage_table(double *dst, const NTT_GLOBAL double *src, uint32_t words, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid * 2; i < words; i += nthreads * 2) {
        const double a = src[i], b = src[i + 1];               // adjacent lanes, adjacent pairs: 16 B per lane either way
        dst[i] = a; dst[i + 1] = b;
    }
}
template <class AR> struct KsMac;
template <> struct KsMac<ArU64> {
    static DEV void mac(uint64_t &acc, uint64_t x, uint64_t key, const DMod &qm, const ArCtx<ArU64> &A) { acc = addmod(acc, mulmod(canon4(x, qm.q), key, qm), qm.q); }
    static DEV void settle(uint64_t (&)[16], const ArCtx<ArU64> &) {}
    static DEV uint64_t sum(uint64_t acc, uint64_t x, const DMod &qm) { return addmod(acc, x, qm.q); }
};
template <int RN> struct KsMac<ArF64T<RN>> {
    typedef ArF64T<RN> ArF64;
    static DEV void mac(double &acc, double x, double key, const DMod &, const ArCtx<ArF64> &A) { acc = __dadd_rn(acc, ArF64::mulmod(x, key, A.m)); }
    static DEV void settle(double (&a)[16], const ArCtx<ArF64> &A) { ArF64::renorm(a, A.m); }
    static DEV double sum(double acc, double x, const DMod &) { return __dadd_rn(acc, x); }
};
// (Measured, not kept: using the N*8 bytes of LDS behind the image for a key prefetch instead of the twiddle table - every wave requests
// the first key component of a digit with global_load_lds_dwordx4 at the start of the digit (global -> LDS without registers, read
// back as ds_read_b128).  Bit-exact, -2 % in the stand-alone loop of tools/ubench_ks.hip, but +4 % in this kernel: 3.78 vs 3.62 ms.
// Likewise s_setprio 3 / 0 for the two waves a SIMD holds, so that their memory waits stop coinciding: -7 % in the stand-alone loop,
// no change here (3.65 vs 3.66 ms) and +30 % on the 100-ciphertext launch.)
#ifndef KS_SGPR_A
#define KS_SGPR_A 1         // FP64 key switch: the first-pass roots of the output limb live in SGPRs for all digits (ArPassA)
#endif
template <class FW0, bool ON> struct KsPassA { typedef FW0 P; };
template <class FW0> struct KsPassA<FW0, true> { typedef ArPassA<FW0> P; };
#ifndef KS_PRE_SYNC
#define KS_PRE_SYNC 1       // the "image is free again" barrier of a digit sits behind the next digit's first pass (ntt_forward_regs<.., PRE>)
#endif
#ifndef KS_MAC_FENCE
#define KS_MAC_FENCE 0      // FP64 path: letting the scheduler interleave key loads with the MACs measured 11-14 % faster (same VGPRs)
#endif
template <int L, class AR, int MINW = 1, bool TWL = false>
__global__ void __launch_bounds__(NttPlan<L>::NT, MINW) k_keyswitch_rr(const uint64_t *__restrict__ target, size_t tgt_stride, const uint64_t *__restrict__ add0,
                                                                 const uint64_t *__restrict__ add1, size_t add_stride, const void *__restrict__ key_,
                                                                 uint64_t *out, const DevConsts *__restrict__ C, int galois, uint32_t accmax,
                                                                 const uint64_t *extra, size_t ex_stride) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t k = C->k, tid = threadIdx.x;
    // (ct, j) with j fastest: the k workgroups of a ciphertext run together and share its source limbs through L2 / MALL.  (A
    // limb-major order that lets an XCD's workgroups share one key slice in L2 was measured: no gain at N = 8192 - the 15.6 MB of
    // keys stream from the infinity cache fast enough - and 30 % slower at N = 16384, where the source limbs then come from HBM
    // once per (limb, half) workgroup.)
    const uint32_t ct = blockIdx.x / k, j = blockIdx.x % k;
    const DMod qm = C->q[j];
    const uint64_t q = qm.q;
    const ArCtx<AR> A(C, j);
    const int dbc = galois ? C->gdbc : C->dbc;
    const uint64_t mask = (1ull << dbc) - 1;
    const size_t kn = (size_t)k * n;
    T acc0[16], acc1[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0; acc1[r] = 0; }
    // TWL: forward twiddles from an LDS copy of the table (FP64 policies, N <= 8192, enough digits per workgroup to pay for staging it)
    typedef typename std::conditional<TWL, typename KsFwd<AR, L>::P, AR>::type FW0;
    typedef typename KsPassA<FW0, KS_SGPR_A && std::is_same<T, double>::value>::P FW;
    typename FW::Tw fwt;
    if constexpr (TWL) {
        static_assert(KsFwd<AR, L>::lds, "LDS twiddles need an FP64 policy and N <= 8192");
        double *tws = reinterpret_cast<double *>(smem) + ntt_lds_words(n);
        stage_table(tws, A.fw.w, n, tid, NttPlan<L>::NT);
        fwt.w = (const __attribute__((address_space(3))) double *)tws;
        __syncthreads();
    } else static_cast<typename FW0::Tw &>(fwt) = A.fw;
    if constexpr (HasPassA<FW>::value) ntt_load_pass_a<SA>(fwt, A.fw.w);
    const T *kp = reinterpret_cast<const T *>(key_);
    uint32_t terms = 0;
    for (uint32_t l = 0; l < k; l++) {
        const uint32_t nd = galois ? C->gk_dig[l] : C->rl_dig[l];
        const uint64_t *src = target + (size_t)ct * tgt_stride + (size_t)l * n;
        uint64_t raw[16];                          // the source words of limb l stay in registers for all of its digits
        {
            uint32_t t0 = tid;
            asm volatile("" : "+v"(t0));
#pragma unroll
            for (int r = 0; r < 16; r++) raw[r] = src[pass_index<L, SA, 0>(t0, r)];
        }
        for (uint32_t d = 0; d < nd; d++, kp += 2 * kn) {
            const int sh = dbc * (int)d;
            uint32_t tl = tid;
            asm volatile("" : "+v"(tl));           // opaque copy of tid: keeps LDS/twiddle address math and twiddle loads inside the
                                                   // loop (hoisted as loop invariants they cost >150 VGPRs and spill)
            T v[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                uint64_t t = (raw[r] >> sh) & mask;
                if constexpr (std::is_same<T, uint64_t>::value) { if (mask >= q) t = t >= q ? bred128(t, 0, qm) : t; }
                v[r] = A.load(t);                  // F64: the first recentring of the transform reduces digits >= q_j
            }
            if constexpr (std::is_same<T, double>::value) { if (mask >= q) AR::renorm(v, A.m); }     // digits below q_j need no recentring (uniform branch)
            ntt_forward_regs<FW, L, KS_PRE_SYNC != 0>(v, s, fwt, A.m, tl);
            const T *k0 = kp + (size_t)j * n, *k1 = kp + kn + (size_t)j * n;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                if (KS_MAC_FENCE || std::is_same<T, uint64_t>::value) __builtin_amdgcn_sched_barrier(0);   // integer path: bounds live key words
                const uint32_t pos = tail_index<L>(tl, r);
                struct alignas(16) P2 { T a, b; };
                const P2 a = *reinterpret_cast<const P2 *>(k0 + pos), b = *reinterpret_cast<const P2 *>(k1 + pos);
                KsMac<AR>::mac(acc0[r], v[r], a.a, qm, A); KsMac<AR>::mac(acc0[r + 1], v[r + 1], a.b, qm, A);
                KsMac<AR>::mac(acc1[r], v[r], b.a, qm, A); KsMac<AR>::mac(acc1[r + 1], v[r + 1], b.b, qm, A);
            }
            if (++terms == accmax) { terms = 0; KsMac<AR>::settle(acc0, A); KsMac<AR>::settle(acc1, A); }
            // LDS of this transform is reused by the next one.  KS_PRE_SYNC: that barrier is inside the next forward transform; the
            // inverse transforms below start with block-local traffic (every wave in its own blocks) or, without NTT_TAIL_LOCAL /
            // for D = 2, still need it here.
            if (!KS_PRE_SYNC) __syncthreads();
        }
    }
    if (KS_PRE_SYNC && !ntt_tail_local<L>()) __syncthreads();
#pragma unroll 1
    for (int p = 0; p < 2; p++) {
        T v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = p ? acc1[r] : acc0[r];
        uint32_t tl = tid;
        asm volatile("" : "+v"(tl));               // see above: no hoisting / sharing of address math across the two transforms
        ntt_inverse_regs<AR, L>(v, s, A.iv, A.m, tl);
        const uint64_t *ad = p ? add1 : add0;
        uint64_t *o = out + ((size_t)ct * 2 + p) * kn + (size_t)j * n;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t e = pass_index<L, SA, 0>(tl, r);
            uint64_t val = A.scaled(v[r]);
            if (ad) val = addmod(val, ad[(size_t)ct * add_stride + (size_t)j * n + e], q);
            if (extra) val = addmod(val, extra[(size_t)ct * ex_stride + (size_t)p * kn + (size_t)j * n + e], q);   // fused "+ accumulator" (may alias out)
            o[e] = val;
        }
        __syncthreads();
    }
}
// N = 16384 key switch without spills.  k_keyswitch_rr<14> needs 1024 threads per limb, which caps a thread at 128 VGPRs - the
// 2 x 16 accumulators + 16 coefficients + twiddles do not fit and go to scratch.  A 2N'-point negacyclic transform is one
// butterfly stage over (i, i + N') followed by two independent N'-point transforms with re-indexed root tables (DevConsts::twdh),
// and the key multiply-accumulate is pointwise - so block = (ct, output limb j, half h) runs the N' = 8192 machinery of
// k_keyswitch_rr<13> (512 threads, 204 VGPRs, no scratch) on its half: stage 0 is folded into the digit load (both inputs of the
// butterfly are read, one output kept), keys are read at h*N' + position.  The two inverse sub-transforms leave through `half`
// and k_ks_combine14 applies the last inverse stage (u + v, (u - v) w^-1), the N^-1 scaling and the (c0, c1) addends.
template <class AR>
__global__ void __launch_bounds__(NttPlan<13>::NT) k_keyswitch_split14(const uint64_t *__restrict__ target, size_t tgt_stride, const void *__restrict__ key_,
                                                                        uint64_t *__restrict__ half, const DevConsts *__restrict__ C, int galois, uint32_t accmax) {
    typedef typename AR::T T;
    static_assert(std::is_same<T, double>::value, "FP64 policies only");
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr int L = 13;
    constexpr uint32_t n2 = 1u << L, n = 2 * n2;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t k = C->k, tid = threadIdx.x;
    const uint32_t h = blockIdx.x & 1, j = (blockIdx.x >> 1) % k, ct = blockIdx.x / (2 * k);
    const DMod qm = C->q[j];
    const ArCtx<AR> A(C, j);
    typedef const NTT_GLOBAL double *GP;
    // (an LDS copy of the half's table, as in k_keyswitch_rr, was measured: -30 % - with one digit per limb a workgroup runs only
    // k = 8 transforms, too few to pay for staging 64 KiB)
    typedef typename KsPassA<AR, KS_SGPR_A != 0>::P FW;
    typename FW::Tw fwh;
    fwh.w = (GP)(C->twdh + ((size_t)(j * 2 + 0) * 2 + h) * n2);
    if constexpr (HasPassA<FW>::value) ntt_load_pass_a<SA>(fwh, fwh.w);
    const typename AR::Tw ivh = {(GP)(C->twdh + ((size_t)(j * 2 + 1) * 2 + h) * n2)};
    const int dbc = galois ? C->gdbc : C->dbc;
    const uint64_t mask = (1ull << dbc) - 1;
    const size_t kn = (size_t)k * n;
    T acc0[16], acc1[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0; acc1[r] = 0; }
    const T *kp = reinterpret_cast<const T *>(key_);
    uint32_t terms = 0;
    for (uint32_t l = 0; l < k; l++) {
        const uint32_t nd = galois ? C->gk_dig[l] : C->rl_dig[l];
        const uint64_t *src = target + (size_t)ct * tgt_stride + (size_t)l * n;
        for (uint32_t d = 0; d < nd; d++, kp += 2 * kn) {
            const int sh = dbc * (int)d;
            uint32_t tl = tid;
            asm volatile("" : "+v"(tl));           // as in k_keyswitch_rr: keep address math and twiddle loads inside the loop
            T v[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t e = pass_index<L, SA, 0>(tl, r);
                T X = A.load((src[e] >> sh) & mask), Y = A.load((src[e + n2] >> sh) & mask);
                AR::fwd(X, Y, A.fw, 1, A.m);        // stage 0 of the 2N'-point transform: (x + w y, x - w y), w = root[1]
                v[r] = h ? Y : X;
            }
            AR::renorm(v, A.m);
            ntt_forward_regs<FW, L, KS_PRE_SYNC != 0>(v, s, fwh, A.m, tl);
            const T *k0 = kp + (size_t)j * n + (size_t)h * n2, *k1 = k0 + kn;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const uint32_t pos = tail_index<L>(tl, r);
                struct alignas(16) P2 { T a, b; };
                const P2 a = *reinterpret_cast<const P2 *>(k0 + pos), b = *reinterpret_cast<const P2 *>(k1 + pos);
                KsMac<AR>::mac(acc0[r], v[r], a.a, qm, A); KsMac<AR>::mac(acc0[r + 1], v[r + 1], a.b, qm, A);
                KsMac<AR>::mac(acc1[r], v[r], b.a, qm, A); KsMac<AR>::mac(acc1[r + 1], v[r + 1], b.b, qm, A);
            }
            if (++terms == accmax) { terms = 0; KsMac<AR>::settle(acc0, A); KsMac<AR>::settle(acc1, A); }
            if (!KS_PRE_SYNC) __syncthreads();
        }
    }
    if (KS_PRE_SYNC && !ntt_tail_local<L>()) __syncthreads();
#pragma unroll 1
    for (int p = 0; p < 2; p++) {
        T v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = p ? acc1[r] : acc0[r];
        uint32_t tl = tid;
        asm volatile("" : "+v"(tl));
        ntt_inverse_regs<AR, L>(v, s, ivh, A.m, tl);
        uint64_t *o = half + ((size_t)ct * 2 + p) * kn + (size_t)j * n + (size_t)h * n2;
#pragma unroll
        for (int r = 0; r < 16; r++) o[pass_index<L, SA, 0>(tl, r)] = A.canon(v[r]);
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_ks_combine14(const uint64_t *__restrict__ half, const uint64_t *__restrict__ add0, const uint64_t *__restrict__ add1,
                                                       size_t add_stride, uint64_t *out, const DevConsts *__restrict__ C, const uint64_t *extra,
                                                       size_t ex_stride) {
    const uint32_t n = C->n, n2 = n >> 1, k = C->k, chunks = n2 / 256;
    const uint32_t i = (blockIdx.x % chunks) * 256 + threadIdx.x, limb = blockIdx.x / chunks;      // limb = (ct*2 + p)*k + j
    const uint32_t j = limb % k, p = (limb / k) & 1, ct = limb / (2 * k);
    const DMod qm = C->q[j];
    const uint64_t *x = half + (size_t)limb * n;
    const uint64_t u = x[i], v = x[i + n2];
    uint64_t lo = mulmod(addmod(u, v, qm.q), C->ninv[j], qm), hi = mulmod(submod(u, v, qm.q), C->ninv_w[j], qm);
    const uint64_t *ad = p ? add1 : add0;
    if (ad) {
        const uint64_t *a = ad + (size_t)ct * add_stride + (size_t)j * n;
        lo = addmod(lo, a[i], qm.q); hi = addmod(hi, a[i + n2], qm.q);
    }
    if (extra) {
        const uint64_t *x2 = extra + (size_t)ct * ex_stride + ((size_t)p * k + j) * n;
        lo = addmod(lo, x2[i], qm.q); hi = addmod(hi, x2[i + n2], qm.q);
    }
    uint64_t *o = out + (size_t)limb * n;
    o[i] = lo; o[i + n2] = hi;
}

// Latency variant of the key switch for SMALL batches (LoLa: one image = 1..13 ciphertexts per rotation): the fused kernel above
// runs count*k workgroups, each pushing all digit transforms through one CU in sequence - 5 busy CUs of 256 at count 1.  Here
// the digit transforms are spread over the chip and the sum is a second launch:
//   k_ks_digit_mac : block = (ct, digit g, output limb j): digit -> forward transform -> times the key pair -> partial products
//                    part[ct][g][2][k][N] (transform order, same 16 B/lane pattern as the key reads)
//   k_ks_sum_intt  : block = (ct, j, component p): sum of the partials over g -> inverse transform -> (+ add_p) -> out
// Same residues as the fused kernel (exact arithmetic in both), HBM traffic 2 * tot * 2kN words per ciphertext more.
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT) k_ks_digit_mac(const uint64_t *__restrict__ target, size_t tgt_stride, const void *__restrict__ key_,
                                                                  void *__restrict__ part_, const DevConsts *__restrict__ C, int galois, uint32_t tot) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t k = C->k, tid = threadIdx.x;
    const uint32_t j = blockIdx.x % k, g = (blockIdx.x / k) % tot, ct = blockIdx.x / (k * tot);
    uint32_t l = 0, d = g;
    for (;; l++) { const uint32_t nd = galois ? C->gk_dig[l] : C->rl_dig[l]; if (d < nd) break; d -= nd; }
    const DMod qm = C->q[j];
    const uint64_t q = qm.q;
    const ArCtx<AR> A(C, j);
    const int dbc = galois ? C->gdbc : C->dbc, sh = dbc * (int)d;
    const uint64_t mask = (1ull << dbc) - 1;
    const size_t kn = (size_t)k * n;
    const uint64_t *src = target + (size_t)ct * tgt_stride + (size_t)l * n;
    T v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        uint64_t t = (src[pass_index<L, SA, 0>(tid, r)] >> sh) & mask;
        if constexpr (std::is_same<T, uint64_t>::value) { if (mask >= q) t = t >= q ? bred128(t, 0, qm) : t; }
        v[r] = A.load(t);
    }
    if constexpr (std::is_same<T, double>::value) { if (mask >= q) AR::renorm(v, A.m); }
    ntt_forward_regs<AR, L>(v, s, A.fw, A.m, tid);
    const T *k0 = reinterpret_cast<const T *>(key_) + (size_t)g * 2 * kn + (size_t)j * n, *k1 = k0 + kn;
    T *o0 = reinterpret_cast<T *>(part_) + (((size_t)ct * tot + g) * 2) * kn + (size_t)j * n, *o1 = o0 + kn;
    struct alignas(16) P2 { T a, b; };
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const uint32_t pos = tail_index<L>(tid, r);
        const P2 a = *reinterpret_cast<const P2 *>(k0 + pos), b = *reinterpret_cast<const P2 *>(k1 + pos);
        P2 x = {0, 0}, y = {0, 0};
        KsMac<AR>::mac(x.a, v[r], a.a, qm, A); KsMac<AR>::mac(x.b, v[r + 1], a.b, qm, A);
        KsMac<AR>::mac(y.a, v[r], b.a, qm, A); KsMac<AR>::mac(y.b, v[r + 1], b.b, qm, A);
        *reinterpret_cast<P2 *>(o0 + pos) = x; *reinterpret_cast<P2 *>(o1 + pos) = y;
    }
}
// Middle ground for batches of ~7-32 ciphertexts: block = (ct, source limb l, output limb j) runs the digits of ONE source limb
// through the fused loop (accumulators in registers) and leaves one partial pair per (ct, l): k*k workgroups per ciphertext
// instead of k (fused) or digits*k (k_ks_digit_mac), and k_ks_sum_intt adds k partials instead of all digits.
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT) k_ks_limb_mac(const uint64_t *__restrict__ target, size_t tgt_stride, const void *__restrict__ key_,
                                                                 void *__restrict__ part_, const DevConsts *__restrict__ C, int galois, uint32_t accmax) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t k = C->k, tid = threadIdx.x;
    const uint32_t j = blockIdx.x % k, l = (blockIdx.x / k) % k, ct = blockIdx.x / (k * k);
    const DMod qm = C->q[j];
    const uint64_t q = qm.q;
    const ArCtx<AR> A(C, j);
    const int dbc = galois ? C->gdbc : C->dbc;
    const uint64_t mask = (1ull << dbc) - 1;
    const size_t kn = (size_t)k * n;
    uint32_t g0 = 0;                                   // index of the first digit of limb l in the key
    for (uint32_t i = 0; i < l; i++) g0 += galois ? C->gk_dig[i] : C->rl_dig[i];
    const uint32_t nd = galois ? C->gk_dig[l] : C->rl_dig[l];
    const uint64_t *src = target + (size_t)ct * tgt_stride + (size_t)l * n;
    uint64_t raw[16];
    {
        uint32_t t0 = tid;
        asm volatile("" : "+v"(t0));
#pragma unroll
        for (int r = 0; r < 16; r++) raw[r] = src[pass_index<L, SA, 0>(t0, r)];
    }
    T acc0[16], acc1[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0; acc1[r] = 0; }
    const T *kp = reinterpret_cast<const T *>(key_) + (size_t)g0 * 2 * kn;
    uint32_t terms = 0;
    for (uint32_t d = 0; d < nd; d++, kp += 2 * kn) {
        const int sh = dbc * (int)d;
        uint32_t tl = tid;
        asm volatile("" : "+v"(tl));
        T v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            uint64_t t = (raw[r] >> sh) & mask;
            if constexpr (std::is_same<T, uint64_t>::value) { if (mask >= q) t = t >= q ? bred128(t, 0, qm) : t; }
            v[r] = A.load(t);
        }
        if constexpr (std::is_same<T, double>::value) { if (mask >= q) AR::renorm(v, A.m); }
        ntt_forward_regs<AR, L, KS_PRE_SYNC != 0>(v, s, A.fw, A.m, tl);
        const T *k0 = kp + (size_t)j * n, *k1 = kp + kn + (size_t)j * n;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const uint32_t pos = tail_index<L>(tl, r);
            struct alignas(16) P2 { T a, b; };
            const P2 a = *reinterpret_cast<const P2 *>(k0 + pos), b = *reinterpret_cast<const P2 *>(k1 + pos);
            KsMac<AR>::mac(acc0[r], v[r], a.a, qm, A); KsMac<AR>::mac(acc0[r + 1], v[r + 1], a.b, qm, A);
            KsMac<AR>::mac(acc1[r], v[r], b.a, qm, A); KsMac<AR>::mac(acc1[r + 1], v[r + 1], b.b, qm, A);
        }
        if (++terms == accmax) { terms = 0; KsMac<AR>::settle(acc0, A); KsMac<AR>::settle(acc1, A); }
        if (!KS_PRE_SYNC) __syncthreads();         // otherwise inside the next forward transform; nothing else uses the image
    }
    KsMac<AR>::settle(acc0, A); KsMac<AR>::settle(acc1, A);        // partials leave recentred: k of them are summed without a check
    T *o0 = reinterpret_cast<T *>(part_) + (((size_t)ct * k + l) * 2) * kn + (size_t)j * n, *o1 = o0 + kn;
    struct alignas(16) P2 { T a, b; };
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const uint32_t pos = tail_index<L>(tid, r);
        *reinterpret_cast<P2 *>(o0 + pos) = P2{acc0[r], acc0[r + 1]};
        *reinterpret_cast<P2 *>(o1 + pos) = P2{acc1[r], acc1[r + 1]};
    }
}
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT) k_ks_sum_intt(const void *__restrict__ part_, const uint64_t *__restrict__ add0, const uint64_t *__restrict__ add1,
                                                                 size_t add_stride, uint64_t *out, const DevConsts *__restrict__ C, uint32_t tot,
                                                                 uint32_t accmax, const uint64_t *extra, size_t ex_stride) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t k = C->k, tid = threadIdx.x;
    const uint32_t p = blockIdx.x & 1, j = (blockIdx.x >> 1) % k, ct = blockIdx.x / (2 * k);
    const DMod qm = C->q[j];
    const ArCtx<AR> A(C, j);
    const size_t kn = (size_t)k * n;
    const T *src = reinterpret_cast<const T *>(part_) + ((size_t)ct * tot * 2 + p) * kn + (size_t)j * n;
    struct alignas(16) P2 { T a, b; };
    T v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = 0;
    uint32_t terms = 0;
    for (uint32_t g = 0; g < tot; g++, src += 2 * kn) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const P2 x = *reinterpret_cast<const P2 *>(src + tail_index<L>(tid, r));
            v[r] = KsMac<AR>::sum(v[r], x.a, qm); v[r + 1] = KsMac<AR>::sum(v[r + 1], x.b, qm);
        }
        if (++terms == accmax) { terms = 0; KsMac<AR>::settle(v, A); }
    }
    ntt_inverse_regs<AR, L>(v, s, A.iv, A.m, tid);
    const uint64_t *ad = p ? add1 : add0;
    uint64_t *o = out + ((size_t)ct * 2 + p) * kn + (size_t)j * n;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t e = pass_index<L, SA, 0>(tid, r);
        uint64_t val = A.scaled(v[r]);
        if (ad) val = addmod(val, ad[(size_t)ct * add_stride + (size_t)j * n + e], qm.q);
        if (extra) val = addmod(val, extra[(size_t)ct * ex_stride + (size_t)p * kn + (size_t)j * n + e], qm.q);
        o[e] = val;
    }
}
// in-place conversion of key words to the FP64 form used by k_keyswitch_rr<L, ArF64> (exact: residues < 2^49)
__global__ void k_u64_to_f64(uint64_t *p, size_t words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) { double d = (double)(long long)p[i]; reinterpret_cast<double *>(p)[i] = d; }
}

// ------------------------------------------------------------------ client-side operations on the device
// KeyGenerator / Encryptor / Decryptor of the data owner, for deployments where the client has a GPU too.  Randomness is a
// counter-based Philox4x32-10 stream keyed by a caller seed: reproducible and statistically sound, NOT a certified DRBG.
struct Philox { uint32_t c[4]; };
DEV Philox philox(uint64_t seed, uint64_t ctr_hi, uint64_t ctr_lo) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {{c0, c1, c2, c3}};
}
// streams: 0 = ternary, 1/2 = noise polys, 3 = uniform
DEV int32_t sample_ternary(uint64_t seed, uint64_t stream, uint64_t item, uint32_t i) {
    for (uint32_t tr = 0;; tr++) {
        Philox p = philox(seed, (stream << 32) | tr, (item << 20) | i);
#pragma unroll
        for (int w = 0; w < 4; w++) for (int b = 0; b < 32; b += 2) { uint32_t v = (p.c[w] >> b) & 3; if (v != 3) return (int32_t)v - 1; }
    }
}
DEV int32_t sample_noise(uint64_t seed, uint64_t stream, uint64_t item, uint32_t i) {      // clipped normal sigma 3.2, 6 sigma, cast
    for (uint32_t tr = 0;; tr++) {
        Philox p = philox(seed, (stream << 32) | tr, (item << 20) | i);
        const double u1 = ((double)(((uint64_t)p.c[0] << 21) ^ (p.c[1] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
        const double u2 = ((double)(((uint64_t)p.c[2] << 21) ^ (p.c[3] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
        const double g = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2) * 3.2;
        if (fabs(g) <= 19.2) return (int32_t)g;
    }
}
DEV uint64_t sample_uniform(uint64_t seed, uint64_t stream, uint64_t item, uint32_t i, uint64_t q) {
    const uint64_t lim = ~0ull - (~0ull % q) - 1;
    for (uint32_t tr = 0;; tr++) {
        Philox p = philox(seed, (stream << 32) | tr, (item << 20) | i);
        uint64_t v = ((uint64_t)p.c[0] << 32) | p.c[1];
        if (v <= lim) return v % q;
        v = ((uint64_t)p.c[2] << 32) | p.c[3];
        if (v <= lim) return v % q;
    }
}
// out[item][j][i]: kind 0 ternary residues, kind 1 noise residues, kind 2 uniform residues mod q_j (item = blockIdx / (k*chunks))
__global__ void k_sample(uint64_t *out, const DevConsts *__restrict__ C, uint32_t chunks, int kind, uint64_t seed, uint64_t stream, uint64_t item0) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t k = C->k, j = limb % k; const uint64_t item = item0 + limb / k, q = C->q[j].q;
    uint64_t v;
    if (kind == 2) v = sample_uniform(seed, stream, item * k + j, i, q);
    else { int32_t s = kind == 0 ? sample_ternary(seed, stream, item, i) : sample_noise(seed, stream, item, i); v = s >= 0 ? (uint64_t)s : q - (uint64_t)(-s); }
    out[(size_t)limb * C->n + i] = v;
}
// b = -(a*s + e) (+ f * snew on limb `hot`), all NTT form; a, e, b: [k][N]; s, snew: [k][N]; f[k] factor per limb
__global__ void k_key_b(const uint64_t *a, const uint64_t *e, const uint64_t *s, const uint64_t *snew, uint64_t factor, int hot, uint64_t *b,
                        const DevConsts *__restrict__ C, uint32_t chunks) {
    uint32_t limb, i; decode(chunks, limb, i);
    const DMod qm = C->q[limb]; size_t o = (size_t)limb * C->n + i;
    uint64_t v = negmod(addmod(mulmod(a[o], s[o], qm), e[o], qm.q), qm.q);
    if ((int)limb == hot) v = addmod(v, mulmod(snew[o], factor, qm), qm.q);
    b[o] = v;
}
__global__ void k_mul_limbs(const uint64_t *a, const uint64_t *b, uint64_t *o, const DevConsts *__restrict__ C, uint32_t chunks) {   // NTT-form product, [k][N]
    uint32_t limb, i; decode(chunks, limb, i);
    size_t x = (size_t)limb * C->n + i; o[x] = mulmod(a[x], b[x], C->q[limb % C->k]);
}
// o[ct][j] = a[ct][j] * b[j] (+ add[ct][j]): b broadcast over ciphertexts, NTT form
__global__ void k_mul_limbs_bcast(const uint64_t *a, const uint64_t *b, const uint64_t *add, uint64_t *o, const DevConsts *__restrict__ C, uint32_t chunks) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t j = limb % C->k; const DMod qm = C->q[j];
    size_t x = (size_t)limb * C->n + i;
    uint64_t v = mulmod(a[x], b[(size_t)j * C->n + i], qm);
    o[x] = add ? addmod(v, add[x], qm.q) : v;
}
// noise probe: acc[ct][j] <- t * (c0[ct][j] + acc[ct][j]) mod q_j  (the polynomial whose centred norm InvariantNoiseBudget measures)
__global__ void k_noise_poly(const uint64_t *__restrict__ c0, size_t ct_stride, uint64_t *__restrict__ acc, const DevConsts *__restrict__ C, uint32_t chunks) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t j = limb % C->k, ct = limb / C->k; const DMod qm = C->q[j];
    const size_t x = (size_t)limb * C->n + i;
    acc[x] = mulmod(addmod(c0[(size_t)ct * ct_stride + (size_t)j * C->n + i], acc[x], qm.q), C->t.q % qm.q, qm);
}
// encryption tail: out[ct][p][j] = INTT(u[ct][j] * pk[p][j]) + e_p (+ Delta*m for p = 0); u in NTT form
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT) k_encrypt_tail(const uint64_t *__restrict__ u, const uint64_t *__restrict__ pk, const uint64_t *__restrict__ pt,
                                                                 uint32_t pt_stride_words, uint64_t *__restrict__ out, const DevConsts *__restrict__ C,
                                                                 uint64_t seed, uint64_t item0) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t k = C->k, tid = threadIdx.x, j = blockIdx.x % k, p = (blockIdx.x / k) & 1, ct = blockIdx.x / (2 * k);
    const ArCtx<AR> A(C, j);
    const TensorOps<AR> ops(C, j);
    const uint64_t *uu = u + ((size_t)ct * k + j) * n, *pp = pk + ((size_t)p * k + j) * n;
    T v[16];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const uint32_t pos = tail_index<L>(tid, r);
        const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(uu + pos), y = *reinterpret_cast<const ulonglong2 *>(pp + pos);
        v[r] = ops.mul(A.load(x.x), A.load(y.x), A); v[r + 1] = ops.mul(A.load(x.y), A.load(y.y), A);
    }
    ntt_inverse_regs<AR, L>(v, s, A.iv, A.m, tid);
    uint64_t *o = out + (((size_t)ct * 2 + p) * k + j) * n;
    const uint64_t q = C->q[j].q;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t e = pass_index<L, SA, 0>(tid, r);
        uint64_t val = A.scaled(v[r]);
        const int32_t ns = sample_noise(seed, 1 + p, item0 + ct, e);
        val = addmod(val, ns >= 0 ? (uint64_t)ns : q - (uint64_t)(-ns), q);
        if (p == 0 && pt) val = addmod(val, scale_plain(C, pt[(size_t)ct * pt_stride_words + e], j), q);
        o[e] = val;
    }
}
// decryption tail: x_j = c0_j + acc_j (coefficient form), then m = round(t*x/q) mod t by the {t, gamma} trick
template <int K>
__global__ void __launch_bounds__(256) k_decrypt_scale(const uint64_t *__restrict__ c0, size_t ct_stride, const uint64_t *__restrict__ acc, uint64_t *__restrict__ plain,
                                                       const DevConsts *__restrict__ C, uint32_t chunks) {
    const uint32_t n = C->n;
    const uint32_t ct = blockIdx.x / chunks, i = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;
    const DMod tm = C->t, gm = C->gamma;
    u128 at = 0, ag = 0;
#pragma unroll
    for (int j = 0; j < K; j++) {
        const DMod qm = C->q[j];
        uint64_t x = addmod(c0[(size_t)ct * ct_stride + (size_t)j * n + i], acc[((size_t)ct * K + j) * n + i], qm.q);
        uint64_t y = mulmod(mulmod(x, C->tg_q[j], qm), C->inv_qhat_q[j], qm);
        at += (u128)y * C->qhat_t[j]; ag += (u128)y * C->qhat_g[j];
    }
    const uint64_t vt = mulmod(bred128(at, tm), C->neg_inv_q_t, tm), vg = mulmod(bred128(ag, gm), C->neg_inv_q_g, gm);
    const uint64_t r = vg > (gm.q >> 1) ? addmod(vt, (gm.q - vg) % tm.q, tm.q) : submod(vt, vg % tm.q, tm.q);
    plain[(size_t)ct * n + i] = r ? mulmod(r, C->inv_g_t, tm) : 0;
}

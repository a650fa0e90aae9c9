// HOT LOOP B: the element-wise BEHZ steps between the batched transforms (base extension q -> Bsk, fast floor + Shenoy-Kumaresan).
// Included by cn_l_behz.hip only.
#pragma once
#include "cn_dev_common.hip.h"

// ------------------------------------------------------------------ HOT LOOP B: BEHZ multiply
// Step 0/1 (fastbconv_mtilde + mont_rq): q -> Bsk, removing q-overflows via m~ = 2^32.
// src ciphertext c = src + c*stride*ctw, or src_tab[c] (deferred per-ciphertext calls: every ciphertext is its own array);
// writes the q copy (for the q-side NTT) and Bsk.
template <int K, int NB = K>
__global__ void __launch_bounds__(256) k_behz_extend(const uint64_t *__restrict__ src, uint32_t stride, const uint64_t *const *__restrict__ src_tab,
                                                     uint64_t *__restrict__ aq, uint64_t *__restrict__ ab, const DevConsts *__restrict__ C, uint32_t chunks) {
    const uint32_t n = C->n;
    const uint32_t cp = blockIdx.x / chunks, i = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;   // cp = ct*2 + poly
    const uint32_t ct = cp >> 1, p = cp & 1;
    const NTT_GLOBAL uint64_t *x = (const NTT_GLOBAL uint64_t *)(src_tab ? src_tab[ct] : src + (size_t)ct * stride * 2 * K * n) + (size_t)p * K * n + i;   // src_tab: one address per ciphertext
    uint64_t *oq = aq + (size_t)cp * K * n + i, *ob = ab + (size_t)cp * (NB + 1) * n + i;
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
    for (int b = 0; b <= NB; b++) {
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
template <int K, int NB = K>
__global__ void __launch_bounds__(256) k_behz_extend_f64(const uint64_t *__restrict__ src, uint32_t stride, const uint64_t *const *__restrict__ src_tab,
                                                         uint64_t *__restrict__ aq, uint64_t *__restrict__ ab, const DevConsts *__restrict__ C, uint32_t chunks) {
    const uint32_t n = C->n;
    const uint32_t cp = blockIdx.x / chunks, i = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;   // cp = ct*2 + poly
    const uint32_t ct = cp >> 1, p = cp & 1;
    const NTT_GLOBAL uint64_t *x = (const NTT_GLOBAL uint64_t *)(src_tab ? src_tab[ct] : src + (size_t)ct * stride * 2 * K * n) + (size_t)p * K * n + i;   // src_tab: one address per ciphertext
    uint64_t *oq = aq + (size_t)cp * K * n + i, *ob = ab + (size_t)cp * (NB + 1) * n + i;
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
    for (int b = 0; b <= NB; b++) {
        const BzF::Mod mb = {C->qd[K + b], C->qinvd[K + b]};
        double acc = BzF::mulmod(r, C->bd.ex_R_bsk[b], mb);
#pragma unroll
        for (int j = 0; j < K; j++) {
            acc = __dadd_rn(acc, BzF::mulmod(y[j], C->bd.ex_Q_bsk[b][j], mb));
            if (K >= 7 && j == 3) acc = BzF::center(acc, mb);          // more than 7 terms of 2.1 b would pass 2^53
        }
        ob[(size_t)b * n] = BzF::to_u64(acc, mb);
    }
}
template <int K, int NB = K>
__global__ void __launch_bounds__(256) k_behz_floor_f64(const uint64_t *__restrict__ dq, const uint64_t *__restrict__ db, uint64_t *__restrict__ out,
                                                        const DevConsts *__restrict__ C, uint32_t chunks) {
    const uint32_t n = C->n;
    const uint32_t cp = blockIdx.x / chunks, i = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;   // cp = ct*3 + poly
    const uint64_t *xq = dq + (size_t)cp * K * n + i, *xb = db + (size_t)cp * (NB + 1) * n + i;
    uint64_t *o = out + (size_t)cp * K * n + i;
    double y[K], f[NB + 1], z[NB];
#pragma unroll
    for (int j = 0; j < K; j++) {
        const BzF::Mod mq = {C->qd[j], C->qinvd[j]};
        y[j] = bz_canon(BzF::mulmod(BzF::from_u64(xq[(size_t)j * n]), C->bd.fl_c1_q[j], mq), mq);     // [x t (q/q_j)^-1]_{q_j}, canonical
    }
#pragma unroll
    for (int b = 0; b <= NB; b++) {
        const BzF::Mod mb = {C->qd[K + b], C->qinvd[K + b]};
        double acc = BzF::mulmod(BzF::from_u64(xb[(size_t)b * n]), C->bd.fl_T_bsk[b], mb);          // (x_b t - conv_b) q^-1, folded
#pragma unroll
        for (int j = 0; j < K; j++) {
            acc = __dadd_rn(acc, BzF::mulmod(y[j], C->bd.fl_N_bsk[b][j], mb));
            if (K >= 7 && j == 3) acc = BzF::center(acc, mb);          // (sums of more than 7 terms of 2.1 b would pass 2^53)
        }
        f[b] = acc;                                                                                    // |f| <= 12.6 b < 2^53
    }
    const BzF::Mod msk = {C->qd[K + NB], C->qinvd[K + NB]};
    double acc = BzF::mulmod(-f[NB], C->bd.inv_B_msk, msk);
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const BzF::Mod mb = {C->qd[K + j], C->qinvd[K + j]};
        z[j] = BzF::mulmod(f[j], C->bd.inv_bhat_b[j], mb);
        acc = __dadd_rn(acc, BzF::mulmod(z[j], C->bd.fl_A_msk[j], msk));
        if (NB >= 7 && j == 3) acc = BzF::center(acc, msk);
    }
    const double alpha = BzF::center(acc, msk);                                                        // centred alpha_sk
#pragma unroll
    for (int j = 0; j < K; j++) {
        const BzF::Mod mq = {C->qd[j], C->qinvd[j]};
        double a2 = BzF::mulmod(-alpha, C->bd.B_q[j], mq);
#pragma unroll
        for (int l = 0; l < NB; l++) {
            a2 = __dadd_rn(a2, BzF::mulmod(z[l], C->bd.bhat_q[j][l], mq));
            if (NB >= 7 && l == 3) a2 = BzF::center(a2, mq);
        }
        o[(size_t)j * n] = BzF::to_u64(a2, mq);
    }
}
// Steps 3/4 (x t, fast_floor: q u Bsk -> Bsk, fastbconv_sk: Bsk -> q) per coefficient.
template <int K, int NB = K>
__global__ void __launch_bounds__(256) k_behz_floor(const uint64_t *__restrict__ dq, const uint64_t *__restrict__ db, uint64_t *__restrict__ out,
                                                    const DevConsts *__restrict__ C, uint32_t chunks) {
    const uint32_t n = C->n;
    const uint32_t cp = blockIdx.x / chunks, i = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;   // cp = ct*3 + poly
    const uint64_t *xq = dq + (size_t)cp * K * n + i, *xb = db + (size_t)cp * (NB + 1) * n + i;
    uint64_t *o = out + (size_t)cp * K * n + i;
    uint64_t y[K], f[NB + 1], z[NB];
#pragma unroll
    for (int j = 0; j < K; j++) y[j] = mulmod(xq[(size_t)j * n], C->fl_c1_q[j], C->q[j]);           // x * t * (q/q_j)^-1
#pragma unroll
    for (int b = 0; b <= NB; b++) {
        const DMod bm = C->bsk[b];
        u128 acc = (u128)xb[(size_t)b * n] * C->fl_T_bsk[b];                                       // (x_b*t - conv_b) * q^-1, folded
#pragma unroll
        for (int j = 0; j < K; j++) acc += (u128)y[j] * C->fl_N_bsk[b][j];
        f[b] = bred128(acc, bm);
    }
    const DMod sk = C->bsk[NB];
    u128 acc = (u128)(sk.q - f[NB]) * C->inv_B_msk;
#pragma unroll
    for (int j = 0; j < NB; j++) { z[j] = mulmod(f[j], C->inv_bhat_b[j], C->bsk[j]); acc += (u128)z[j] * C->fl_A_msk[j]; }
    const uint64_t alpha = bred128(acc, sk);
    const bool neg = alpha > (sk.q >> 1);
#pragma unroll
    for (int j = 0; j < K; j++) {
        const DMod qm = C->q[j];
        u128 a2 = neg ? (u128)C->B_q[j] * (sk.q - alpha) : (u128)(qm.q - C->B_q[j]) * alpha;        // -alpha*B (centred alpha)
#pragma unroll
        for (int l = 0; l < NB; l++) a2 += (u128)z[l] * C->bhat_q[j][l];
        o[(size_t)j * n] = bred128(a2, qm);
    }
}

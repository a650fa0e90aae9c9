// Register-radix negacyclic NTT core for gfx950 (wave64).
//
// One workgroup of NT = N/16 threads owns one limb; every thread keeps 16 coefficients (32 VGPRs) and runs up to four
// butterfly stages on them without touching memory.  The log2(N) stages are split as  [SA, 4, 4 | D]  (SA = log2N-8-D,
// D = 1, or 2 for N = 16384):  pass A reads its operands straight from global memory (stride N/2^SA between a thread's
// operands, consecutive lanes -> consecutive addresses: coalesced), passes B and C exchange through LDS, and pass D works
// on 2^D adjacent coefficients per thread so the results leave as 16 B/lane fully coalesced stores.  Three LDS exchanges
// and three barriers per transform instead of log2(N) of each.  The inverse transform mirrors the same passes.
//
// LDS image: coefficient e lives at P(e) = e + 2*(e >> 5) (two u64 of padding per 32): every exchange pattern below is then
// bank-conflict free for ds_read/write_b64 (half-wave groups) and ds_read/write_b128.
//
// Butterflies are Harvey lazy (values in [0,4q) forward, [0,2q) inverse) with Shoup twiddle companions; tables are the
// bit-reversed ones of cn_tables.cpp, so the output order equals SEAL's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NTT_DEV __device__ __forceinline__

NTT_DEV uint32_t lds_pos(uint32_t e) { return e + 2u * (e >> 5); }
__host__ __device__ inline uint32_t ntt_lds_words(uint32_t n) { return n + 2u * (n >> 5); }

NTT_DEV uint64_t ntt_mulshoup(uint64_t y, uint64_t w, uint64_t ws, uint64_t q) { return y * w - __umul64hi(ws, y) * q; }

// forward butterfly (Cooley-Tukey): (X, Y) -> (X + W*Y, X - W*Y), lazily in [0,4q)
NTT_DEV void bfly_fwd(uint64_t &X, uint64_t &Y, uint64_t W, uint64_t Ws, uint64_t q, uint64_t q2) {
    uint64_t x = X - ((X >= q2) ? q2 : 0);
    uint64_t t = ntt_mulshoup(Y, W, Ws, q);
    X = x + t;
    Y = x + q2 - t;
}
// inverse butterfly (Gentleman-Sande): (U, V) -> (U + V, (U - V) * W), both in [0,2q)
NTT_DEV void bfly_inv(uint64_t &U, uint64_t &V, uint64_t W, uint64_t Ws, uint64_t q, uint64_t q2) {
    uint64_t s = U + V;
    uint64_t d = U + q2 - V;
    U = s - ((s >= q2) ? q2 : 0);
    V = ntt_mulshoup(d, W, Ws, q);
}

template <int L> struct NttPlan {
    static constexpr int D = (L == 14) ? 2 : 1;      // stages of the last (adjacent-coefficient) pass
    static constexpr int SA = L - 8 - D;             // stages of the first pass (1..4)
    static constexpr int NT = 1 << (L - 4);          // threads per limb
    static_assert(SA >= 1 && SA <= 4, "unsupported transform size");
};

// coefficient index of register r of thread tid in a pass covering stages [S0, S0+S)
template <int L, int S, int S0> NTT_DEV uint32_t pass_index(uint32_t tid, int r) {
    constexpr int NT = 1 << (L - 4), LO = L - S0 - S;
    const uint32_t G = (uint32_t)(r >> S) * NT + tid, mid = r & ((1 << S) - 1);
    const uint32_t lo = G & ((1u << LO) - 1), hi = G >> LO;
    return (hi << (L - S0)) | (mid << LO) | lo;
}
template <int L, int S, int S0> NTT_DEV uint32_t pass_hi(uint32_t tid, int g) {
    constexpr int NT = 1 << (L - 4), LO = L - S0 - S;
    return ((uint32_t)g * NT + tid) >> LO;
}

// S butterfly stages on the 16 registers (16 >> S independent groups of 2^S coefficients)
template <int L, int S, int S0> NTT_DEV void fwd_stages(uint64_t (&x)[16], const uint64_t *__restrict__ w, const uint64_t *__restrict__ ws, uint64_t q, uint32_t tid) {
    const uint64_t q2 = 2 * q;
#pragma unroll
    for (int u = 0; u < S; u++) {
        __builtin_amdgcn_sched_barrier(0);            // one stage's twiddles live at a time (register pressure)
        const int half = 1 << (S - 1 - u);
#pragma unroll
        for (int g = 0; g < (16 >> S); g++) {
            const uint32_t hi = pass_hi<L, S, S0>(tid, g);
#pragma unroll
            for (int blk = 0; blk < (1 << u); blk++) {
                const uint32_t ti = (1u << (S0 + u)) + ((hi << u) | (uint32_t)blk);
                const uint64_t W = w[ti], Ws = ws[ti];
#pragma unroll
                for (int j = 0; j < half; j++) {
                    const int a = (g << S) + blk * 2 * half + j;
                    bfly_fwd(x[a], x[a + half], W, Ws, q, q2);
                }
            }
        }
    }
}
template <int L, int S, int S0> NTT_DEV void inv_stages(uint64_t (&x)[16], const uint64_t *__restrict__ iw, const uint64_t *__restrict__ iws, uint64_t q, uint32_t tid) {
    const uint64_t q2 = 2 * q;
#pragma unroll
    for (int u = S - 1; u >= 0; u--) {
        __builtin_amdgcn_sched_barrier(0);
        const int half = 1 << (S - 1 - u);
#pragma unroll
        for (int g = 0; g < (16 >> S); g++) {
            const uint32_t hi = pass_hi<L, S, S0>(tid, g);
#pragma unroll
            for (int blk = 0; blk < (1 << u); blk++) {
                const uint32_t ti = (1u << (S0 + u)) + ((hi << u) | (uint32_t)blk);
                const uint64_t W = iw[ti], Ws = iws[ti];
#pragma unroll
                for (int j = 0; j < half; j++) {
                    const int a = (g << S) + blk * 2 * half + j;
                    bfly_inv(x[a], x[a + half], W, Ws, q, q2);
                }
            }
        }
    }
}
template <int L, int S, int S0> NTT_DEV void lds_put(const uint64_t (&x)[16], uint64_t *s, uint32_t tid) {
#pragma unroll
    for (int r = 0; r < 16; r++) s[lds_pos(pass_index<L, S, S0>(tid, r))] = x[r];
}
template <int L, int S, int S0> NTT_DEV void lds_get(uint64_t (&x)[16], const uint64_t *s, uint32_t tid) {
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = s[lds_pos(pass_index<L, S, S0>(tid, r))];
}
// last pass layout: register r holds coefficient  (c << D) + (r & (2^D - 1)),  c = tid + NT * (r >> D)
template <int L> NTT_DEV uint32_t tail_index(uint32_t tid, int r) {
    constexpr int D = NttPlan<L>::D, NT = NttPlan<L>::NT;
    return ((tid + (uint32_t)NT * (uint32_t)(r >> D)) << D) + (uint32_t)(r & ((1 << D) - 1));
}
template <int L> NTT_DEV void lds_put_tail(const uint64_t (&x)[16], uint64_t *s, uint32_t tid) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        ulonglong2 v; v.x = x[r]; v.y = x[r + 1];
        *reinterpret_cast<ulonglong2 *>(s + lds_pos(tail_index<L>(tid, r))) = v;
    }
}
template <int L> NTT_DEV void lds_get_tail(uint64_t (&x)[16], const uint64_t *s, uint32_t tid) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(s + lds_pos(tail_index<L>(tid, r)));
        x[r] = v.x; x[r + 1] = v.y;
    }
}
template <int L> NTT_DEV void fwd_tail(uint64_t (&x)[16], const uint64_t *__restrict__ w, const uint64_t *__restrict__ ws, uint64_t q, uint32_t tid) {
    constexpr int D = NttPlan<L>::D, NT = NttPlan<L>::NT;
    const uint64_t q2 = 2 * q;
    if (D == 2) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t c = tid + (uint32_t)NT * g, t0 = (1u << (L - 2)) + c;
            const uint64_t W = w[t0], Ws = ws[t0];
            bfly_fwd(x[4 * g], x[4 * g + 2], W, Ws, q, q2);
            bfly_fwd(x[4 * g + 1], x[4 * g + 3], W, Ws, q, q2);
        }
    }
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const uint32_t pair = tail_index<L>(tid, 2 * p) >> 1, ti = (1u << (L - 1)) + pair;
        bfly_fwd(x[2 * p], x[2 * p + 1], w[ti], ws[ti], q, q2);
    }
}
template <int L> NTT_DEV void inv_tail(uint64_t (&x)[16], const uint64_t *__restrict__ iw, const uint64_t *__restrict__ iws, uint64_t q, uint32_t tid) {
    constexpr int D = NttPlan<L>::D, NT = NttPlan<L>::NT;
    const uint64_t q2 = 2 * q;
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const uint32_t pair = tail_index<L>(tid, 2 * p) >> 1, ti = (1u << (L - 1)) + pair;
        bfly_inv(x[2 * p], x[2 * p + 1], iw[ti], iws[ti], q, q2);
    }
    if (D == 2) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t c = tid + (uint32_t)NT * g, t0 = (1u << (L - 2)) + c;
            const uint64_t W = iw[t0], Ws = iws[t0];
            bfly_inv(x[4 * g], x[4 * g + 2], W, Ws, q, q2);
            bfly_inv(x[4 * g + 1], x[4 * g + 3], W, Ws, q, q2);
        }
    }
}

// Forward transform.  In: x[r] = coefficient pass_index<L,SA,0>(tid,r) (canonical or < 4q).  Out: x[r] = value at
// bit-reversed position tail_index<L>(tid,r), lazily in [0,4q).  `s` = LDS scratch of ntt_lds_words(N) u64.
template <int L> NTT_DEV void ntt_forward_regs(uint64_t (&x)[16], uint64_t *s, const uint64_t *__restrict__ w, const uint64_t *__restrict__ ws, uint64_t q, uint32_t tid) {
    constexpr int SA = NttPlan<L>::SA;
    fwd_stages<L, SA, 0>(x, w, ws, q, tid);
    lds_put<L, SA, 0>(x, s, tid);
    __syncthreads();
    lds_get<L, 4, SA>(x, s, tid);
    fwd_stages<L, 4, SA>(x, w, ws, q, tid);
    lds_put<L, 4, SA>(x, s, tid);
    __syncthreads();
    lds_get<L, 4, SA + 4>(x, s, tid);
    fwd_stages<L, 4, SA + 4>(x, w, ws, q, tid);
    lds_put<L, 4, SA + 4>(x, s, tid);
    __syncthreads();
    lds_get_tail<L>(x, s, tid);
    fwd_tail<L>(x, w, ws, q, tid);
}
// Inverse transform (without the 1/N factor).  In: x[r] = value at position tail_index<L>(tid,r), in [0,2q).
// Out: x[r] = coefficient pass_index<L,SA,0>(tid,r), in [0,2q).
template <int L> NTT_DEV void ntt_inverse_regs(uint64_t (&x)[16], uint64_t *s, const uint64_t *__restrict__ iw, const uint64_t *__restrict__ iws, uint64_t q, uint32_t tid) {
    constexpr int SA = NttPlan<L>::SA;
    inv_tail<L>(x, iw, iws, q, tid);
    lds_put_tail<L>(x, s, tid);
    __syncthreads();
    lds_get<L, 4, SA + 4>(x, s, tid);
    inv_stages<L, 4, SA + 4>(x, iw, iws, q, tid);
    lds_put<L, 4, SA + 4>(x, s, tid);
    __syncthreads();
    lds_get<L, 4, SA>(x, s, tid);
    inv_stages<L, 4, SA>(x, iw, iws, q, tid);
    lds_put<L, 4, SA>(x, s, tid);
    __syncthreads();
    lds_get<L, SA, 0>(x, s, tid);
    inv_stages<L, SA, 0>(x, iw, iws, q, tid);
}

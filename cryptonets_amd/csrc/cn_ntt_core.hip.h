// Register-radix negacyclic NTT core for gfx950 (wave64).
//
// One workgroup of NT = N/16 threads owns one limb; every thread keeps 16 coefficients (32 VGPRs) and runs up to four
// butterfly stages on them without touching memory.  The log2(N) stages are split as  [SA, 4, 4 | D]  (SA = log2N-8-D,
// D = 1, or 2 for N = 16384):  pass A reads its operands straight from global memory (stride N/2^SA between a thread's
// operands, consecutive lanes -> consecutive addresses: coalesced), passes B and C exchange through LDS, and pass D works
// on 2^D adjacent coefficients per thread so the results leave as 16 B/lane fully coalesced stores.  Three LDS exchanges
// per transform instead of log2(N), and only the first one (the transpose behind pass A) needs a workgroup barrier: after
// pass A the transform falls apart into 512-coefficient blocks that one half-wave owns through passes B, C and (D = 1) the
// last pass, so those exchanges are ordered by the wave's own LDS queue.  The inverse transform mirrors the same passes.
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
#define NTT_GLOBAL __attribute__((address_space(1)))      // tables are in global memory: global_load, not flat_load
#ifndef NTT_STAGE_FENCE
#define NTT_STAGE_FENCE 0
#endif
#ifndef NTT_FWD_RECENTRE_ALL
#define NTT_FWD_RECENTRE_ALL 0      // 1: the forward FP64 transform of moduli above 44 bits recentres before every pass (rounds 1-4; A/B)
#endif
#ifndef NTT_TAIL_LOCAL
#define NTT_TAIL_LOCAL 1      // D = 1: the last pass stays inside the half-wave that owns a 512-coefficient block (see tail_index)
#endif

NTT_DEV uint32_t lds_pos(uint32_t e) { return e + 2u * (e >> 5); }
__host__ __device__ constexpr inline uint32_t ntt_lds_words(uint32_t n) { return n + 2u * (n >> 5); }

NTT_DEV uint64_t ntt_mulshoup(uint64_t y, uint64_t w, uint64_t ws, uint64_t q) { return y * w - __umul64hi(ws, y) * q; }

// ---- arithmetic policy 1: unsigned 64-bit, Harvey lazy butterflies with Shoup twiddles (any modulus < 2^61)
struct ArU64 {
    typedef uint64_t T;
    struct Mod { uint64_t q, q2; };
    struct Tw { const NTT_GLOBAL uint64_t *w; const NTT_GLOBAL uint64_t *ws; };        // forward or inverse table pair
    // (X, Y) -> (X + W*Y, X - W*Y), values lazily in [0,4q)
    static NTT_DEV void fwd(T &X, T &Y, const Tw &t, uint32_t ti, const Mod &m) {
        const uint64_t W = t.w[ti], Ws = t.ws[ti];
        uint64_t x = X - ((X >= m.q2) ? m.q2 : 0);
        uint64_t p = ntt_mulshoup(Y, W, Ws, m.q);
        X = x + p;
        Y = x + m.q2 - p;
    }
    // (U, V) -> (U + V, (U - V) * W), values in [0,2q)
    static NTT_DEV void inv(T &U, T &V, const Tw &t, uint32_t ti, const Mod &m) {
        const uint64_t W = t.w[ti], Ws = t.ws[ti];
        uint64_t s = U + V, d = U + m.q2 - V;
        U = s - ((s >= m.q2) ? m.q2 : 0);
        V = ntt_mulshoup(d, W, Ws, m.q);
    }
    template <int SITE> static NTT_DEV void renorm_at(T (&)[16], const Mod &) {}
    static NTT_DEV void renorm(T (&)[16], const Mod &) {}
};

// ---- arithmetic policy 2: exact FP64 for moduli below 2^49.4.  Coefficients are doubles holding exact (signed, lazy)
// integers |x| < 2^53.  w*y mod q: p = w*y rounded, e = fma(w,y,-p) its exact error, h = rint(p/q) (|h - wy/q| <= 2),
// r = fma(-h, q, p) + e is EXACT (|r| <= 2q).  v_fma_f64 is half rate like v_mad_u64_u32 but yields a 53-bit product, so a
// butterfly is 8 FP64 instructions instead of ~60 integer ones; results are bit-identical after canonicalisation.
// recentring sites of a transform
enum { RS_FWD_B = 0, RS_INV_START = 1, RS_INV_C = 2, RS_INV_B = 3, RS_INV_A = 4, RS_FWD_C = 5, RS_FWD_TAIL = 6 };

// RN = 0: moduli <= 44 bits have 2^9 of head-room - the forward transform never recentres (|x| <= q + 13*2.1q = 28.3q < 2^49), the
// inverse only where the doubling sum path needs it.
// RN = 1: moduli below 2^49 (cn_build_f64_tables admits nothing larger).  A forward butterfly is X +- r with r = mulmod(Y, w):
// |h - Yw/q| <= 1/2 + |Yw/q| 3 2^-53 (three roundings: the product, 1/q, their product), so for |Y| <= B q and w < q < 2^49
// |r| <= (1/2 + 0.1875 B) q and a stage takes the bound b (in units of q) to 1.1875 b + 1/2: from canonical input (b = 1) the eight stages
// of the first two passes reach 11.83 q < 2^53 / q = 16 q, from a recentred image (b = 1/2) the six stages of the last two 6.22 q.  So the
// forward transform recentres ONCE, in front of the third pass (round 5; before every pass until then: 2 x 48 FP64 instructions of ~1400
// per transform more); its output is |x| <= 6.3 q (consumers multiply by a canonical word or recentre).  The inverse doubles its sum path
// every stage and keeps recentring before every pass.
template <int RN> struct ArF64T {
    typedef double T;
    struct Mod { double q, qinv; };
    struct Tw { const NTT_GLOBAL double *w; };
    static NTT_DEV double mulmod(double y, double w, const Mod &m) {
        const double p = __dmul_rn(y, w);
        const double e = __fma_rn(y, w, -p);
        const double h = __builtin_rint(__dmul_rn(p, m.qinv));
        return __dadd_rn(__fma_rn(-h, m.q, p), e);
    }
    static NTT_DEV double center(double x, const Mod &m) {          // |result| <= q/2 (+1 ulp of the quotient)
        return __fma_rn(-__builtin_rint(__dmul_rn(x, m.qinv)), m.q, x);
    }
    static NTT_DEV void fwd(T &X, T &Y, const Tw &t, uint32_t ti, const Mod &m) {
        const double p = mulmod(Y, t.w[ti], m);
        Y = __dadd_rn(X, -p);
        X = __dadd_rn(X, p);
    }
    static NTT_DEV void inv(T &U, T &V, const Tw &t, uint32_t ti, const Mod &m) {
        const double s = __dadd_rn(U, V), d = __dadd_rn(U, -V);
        U = s;
        V = mulmod(d, t.w[ti], m);
    }
    // growth bound: a pass of <= 4 stages grows |x| from q/2 to <= 8.5q < 2^53 for q < 2^49.4 -> recentre once per pass
    static NTT_DEV void renorm(T (&x)[16], const Mod &m) {
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = center(x[r], m);
    }
    template <int SITE> static NTT_DEV void renorm_at(T (&x)[16], const Mod &m) {
        if (SITE == RS_FWD_B || SITE == RS_FWD_TAIL) { if (NTT_FWD_RECENTRE_ALL && RN == 1) renorm(x, m); }
        else if (RN == 1 || SITE == RS_INV_START || SITE == RS_INV_B || SITE == RS_INV_A) renorm(x, m);
    }
    static NTT_DEV uint64_t to_u64(double x, const Mod &m) {        // canonical residue of any |x| < 2^53
        double r = center(x, m);
        r = r < 0.0 ? __dadd_rn(r, m.q) : r;
        // exact integer 0 <= r < 2^52: adding 2^52 leaves r in the mantissa bits
        return (uint64_t)__double_as_longlong(__dadd_rn(r, 4503599627370496.0)) & 0x000FFFFFFFFFFFFFull;
    }
    static NTT_DEV double from_u64(uint64_t v) {                    // v < 2^52: build 2^52 + v bitwise, subtract 2^52
        return __dadd_rn(__longlong_as_double((long long)(v | 0x4330000000000000ull)), -4503599627370496.0);
    }
};
typedef ArF64T<1> ArF64;
typedef ArF64T<0> ArF64L;
// The same arithmetic with the twiddle table in LDS: a kernel that pushes many transforms of ONE modulus through a workgroup (the key
// switch: 25 digits per output limb) copies the 8 N-byte table next to the exchange image once; a twiddle is then a ds_read (~64
// cycles) instead of a global load from L2 (~600 cycles, issued after every barrier of the transform with 2 waves per SIMD to hide it).
template <int RN> struct ArF64LdsT : ArF64T<RN> {
    typedef typename ArF64T<RN>::T T;
    typedef typename ArF64T<RN>::Mod Mod;
    struct Tw { const __attribute__((address_space(3))) double *w; };
    static NTT_DEV void fwd(T &X, T &Y, const Tw &t, uint32_t ti, const Mod &m) {
        const double p = ArF64T<RN>::mulmod(Y, t.w[ti], m);
        Y = __dadd_rn(X, -p);
        X = __dadd_rn(X, p);
    }
    static NTT_DEV void inv(T &U, T &V, const Tw &t, uint32_t ti, const Mod &m) {
        const double s = __dadd_rn(U, V), d = __dadd_rn(U, -V);
        U = s;
        V = ArF64T<RN>::mulmod(d, t.w[ti], m);
    }
};

// First-pass twiddles in SGPRs.  In the first pass (stages 0..SA-1) every thread of the workgroup uses the SAME roots - stage u, block
// blk: roots[2^u + blk], 2^SA - 1 <= 15 values per modulus (pass_hi is 0 there: g*NT + tid < 2^(L-SA)).  A kernel that pushes many
// transforms of one modulus through a workgroup loads them once (ntt_load_pass_a) and the butterflies take them as scalar operands:
// no LDS / L2 read and no VGPRs for a quarter of the butterflies.  BASE = ArF64T<RN> or ArF64LdsT<RN>.
NTT_DEV double ntt_uniform(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
template <class BASE> struct ArPassA : BASE {
    typedef typename BASE::T T;
    typedef typename BASE::Mod Mod;
    struct Tw : BASE::Tw { double wa[16]; };
    static constexpr bool passA = true;
    static NTT_DEV void fwdA(T &X, T &Y, const Tw &t, int ti, const Mod &m) {
        const double p = BASE::mulmod(Y, t.wa[ti], m);
        Y = __dadd_rn(X, -p);
        X = __dadd_rn(X, p);
    }
    static NTT_DEV void invA(T &U, T &V, const Tw &t, int ti, const Mod &m) {
        const double s = __dadd_rn(U, V), d = __dadd_rn(U, -V);
        U = s;
        V = BASE::mulmod(d, t.wa[ti], m);
    }
};
template <class AR, class = void> struct HasPassA { static constexpr bool value = false; };
template <class AR> struct HasPassA<AR, decltype((void)AR::passA)> { static constexpr bool value = true; };
// wa[1 .. 2^SA) <- table[1 .. 2^SA) of a GLOBAL root table (uniform addresses), forced into SGPRs
template <int SA, class TW> NTT_DEV void ntt_load_pass_a(TW &t, const NTT_GLOBAL double *table) {
#pragma unroll
    for (int i = 1; i < (1 << SA); i++) t.wa[i] = ntt_uniform(table[i]);
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
template <class AR, int L, int S, int S0> NTT_DEV void fwd_stages(typename AR::T (&x)[16], const typename AR::Tw &tw, const typename AR::Mod &m, uint32_t tid) {
#pragma unroll
    for (int u = 0; u < S; u++) {
        if (NTT_STAGE_FENCE) __builtin_amdgcn_sched_barrier(0);            // one stage's twiddles live at a time (register pressure)
        const int half = 1 << (S - 1 - u);
#pragma unroll
        for (int g = 0; g < (16 >> S); g++) {
            const uint32_t hi = S0 == 0 ? 0u : pass_hi<L, S, S0>(tid, g);          // first pass: g*NT + tid < 2^(L-S), same roots for all threads
#pragma unroll
            for (int blk = 0; blk < (1 << u); blk++) {
                const uint32_t ti = (1u << (S0 + u)) + ((hi << u) | (uint32_t)blk);
#pragma unroll
                for (int j = 0; j < half; j++) {
                    const int a = (g << S) + blk * 2 * half + j;
                    if constexpr (S0 == 0 && HasPassA<AR>::value) AR::fwdA(x[a], x[a + half], tw, (1 << u) + blk, m);
                    else AR::fwd(x[a], x[a + half], tw, ti, m);
                }
            }
        }
    }
}
template <class AR, int L, int S, int S0> NTT_DEV void inv_stages(typename AR::T (&x)[16], const typename AR::Tw &tw, const typename AR::Mod &m, uint32_t tid) {
#pragma unroll
    for (int u = S - 1; u >= 0; u--) {
        if (NTT_STAGE_FENCE) __builtin_amdgcn_sched_barrier(0);
        const int half = 1 << (S - 1 - u);
#pragma unroll
        for (int g = 0; g < (16 >> S); g++) {
            const uint32_t hi = S0 == 0 ? 0u : pass_hi<L, S, S0>(tid, g);          // first pass: g*NT + tid < 2^(L-S), same roots for all threads
#pragma unroll
            for (int blk = 0; blk < (1 << u); blk++) {
                const uint32_t ti = (1u << (S0 + u)) + ((hi << u) | (uint32_t)blk);
#pragma unroll
                for (int j = 0; j < half; j++) {
                    const int a = (g << S) + blk * 2 * half + j;
                    if constexpr (S0 == 0 && HasPassA<AR>::value) AR::invA(x[a], x[a + half], tw, (1 << u) + blk, m);
                    else AR::inv(x[a], x[a + half], tw, ti, m);
                }
            }
        }
    }
}
template <class T, int L, int S, int S0> NTT_DEV void lds_put(const T (&x)[16], T *s, uint32_t tid) {
#pragma unroll
    for (int r = 0; r < 16; r++) s[lds_pos(pass_index<L, S, S0>(tid, r))] = x[r];
}
template <class T, int L, int S, int S0> NTT_DEV void lds_get(T (&x)[16], const T *s, uint32_t tid) {
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = s[lds_pos(pass_index<L, S, S0>(tid, r))];
}
// last pass layout.  D = 2 (and NTT_TAIL_LOCAL = 0): register r holds coefficient  (c << D) + (r & (2^D - 1)),  c = tid + NT * (r >> D)
// - consecutive lanes own consecutive 16 B across the whole limb, which needs a workgroup exchange.
// D = 1, NTT_TAIL_LOCAL: after pass C the 2^9-coefficient block b = tid >> 5 is held by the 32 threads 32b..32b+31 (half a wave), and
// the last stage only pairs neighbours - so thread u = tid & 31 takes the pairs u + 32*(r >> 1) OF ITS OWN BLOCK: coefficient
// 512 b + 2 (u + 32 (r >> 1)) + (r & 1).  A half-wave still covers 512 contiguous bytes per register pair (global stores / key loads
// stay fully coalesced, the LDS image is read as conflict-free ds_read_b128), and the exchange C -> tail needs no workgroup barrier.
template <int L> constexpr bool ntt_tail_local() { return NTT_TAIL_LOCAL && NttPlan<L>::D == 1; }
template <int L> NTT_DEV uint32_t tail_index(uint32_t tid, int r) {
    constexpr int D = NttPlan<L>::D, NT = NttPlan<L>::NT;
    if (ntt_tail_local<L>())
        return ((tid >> 5) << 9) + (((tid & 31u) + 32u * (uint32_t)(r >> 1)) << 1) + (uint32_t)(r & 1);
    return ((tid + (uint32_t)NT * (uint32_t)(r >> D)) << D) + (uint32_t)(r & ((1 << D) - 1));
}
template <class T, int L> NTT_DEV void lds_put_tail(const T (&x)[16], T *s, uint32_t tid) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        struct alignas(16) P2 { T a, b; } v{x[r], x[r + 1]};
        *reinterpret_cast<P2 *>(s + lds_pos(tail_index<L>(tid, r))) = v;
    }
}
template <class T, int L> NTT_DEV void lds_get_tail(T (&x)[16], const T *s, uint32_t tid) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        struct alignas(16) P2 { T a, b; };
        const P2 v = *reinterpret_cast<const P2 *>(s + lds_pos(tail_index<L>(tid, r)));
        x[r] = v.a; x[r + 1] = v.b;
    }
}
template <class AR, int L> NTT_DEV void fwd_tail(typename AR::T (&x)[16], const typename AR::Tw &tw, const typename AR::Mod &m, uint32_t tid) {
    constexpr int D = NttPlan<L>::D, NT = NttPlan<L>::NT;
    if (D == 2) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t c = tid + (uint32_t)NT * g, t0 = (1u << (L - 2)) + c;
            AR::fwd(x[4 * g], x[4 * g + 2], tw, t0, m);
            AR::fwd(x[4 * g + 1], x[4 * g + 3], tw, t0, m);
        }
    }
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const uint32_t pair = tail_index<L>(tid, 2 * p) >> 1, ti = (1u << (L - 1)) + pair;
        AR::fwd(x[2 * p], x[2 * p + 1], tw, ti, m);
    }
}
template <class AR, int L> NTT_DEV void inv_tail(typename AR::T (&x)[16], const typename AR::Tw &tw, const typename AR::Mod &m, uint32_t tid) {
    constexpr int D = NttPlan<L>::D, NT = NttPlan<L>::NT;
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const uint32_t pair = tail_index<L>(tid, 2 * p) >> 1, ti = (1u << (L - 1)) + pair;
        AR::inv(x[2 * p], x[2 * p + 1], tw, ti, m);
    }
    if (D == 2) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t c = tid + (uint32_t)NT * g, t0 = (1u << (L - 2)) + c;
            AR::inv(x[4 * g], x[4 * g + 2], tw, t0, m);
            AR::inv(x[4 * g + 1], x[4 * g + 3], tw, t0, m);
        }
    }
}

// The exchange between the two middle passes never leaves a wave: after pass A the transform splits into independent blocks of
// 2^(8+D) coefficients, and passes B and C of such a block are both owned by the same 2^(4+D) <= 64 consecutive threads.  LDS
// operations of one wave execute in order, so that exchange needs no workgroup barrier - only a scheduling fence.  (One of three
// s_barrier per transform removed; with 2 waves per SIMD that share every barrier, barrier stalls are the main idle time.)
NTT_DEV void ntt_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Forward transform.  In: x[r] = coefficient pass_index<L,SA,0>(tid,r) (canonical).  Out: x[r] = value at bit-reversed
// position tail_index<L>(tid,r), lazy (U64: [0,4q); F64: |x| <= 6.3q, see ArF64T).  `s` = LDS scratch of ntt_lds_words(N) elements.
// PRE: a kernel that pushes one transform after another through the same image (key switch) passes PRE = true instead of ending
// every iteration with a barrier: the barrier "everybody has finished reading the previous image" then sits AFTER the first pass'
// arithmetic, where stragglers have 256 FP64 instructions of slack, and the barrier behind the put finds the waves already aligned.
template <class AR, int L, bool PRE = false> NTT_DEV void ntt_forward_regs(typename AR::T (&x)[16], typename AR::T *s, const typename AR::Tw &tw, const typename AR::Mod &m, uint32_t tid) {
    typedef typename AR::T T;
    constexpr int SA = NttPlan<L>::SA;
    fwd_stages<AR, L, SA, 0>(x, tw, m, tid);
    if (PRE) __syncthreads();
    lds_put<T, L, SA, 0>(x, s, tid);
    __syncthreads();
    lds_get<T, L, 4, SA>(x, s, tid);
    AR::template renorm_at<RS_FWD_B>(x, m);
    fwd_stages<AR, L, 4, SA>(x, tw, m, tid);
    lds_put<T, L, 4, SA>(x, s, tid);
    ntt_wave_sync();                      // pass B -> C stays inside the 2^(4+D) threads that own one 2^(8+D)-coefficient block
    lds_get<T, L, 4, SA + 4>(x, s, tid);
    AR::template renorm_at<RS_FWD_C>(x, m);
    fwd_stages<AR, L, 4, SA + 4>(x, tw, m, tid);
    lds_put<T, L, 4, SA + 4>(x, s, tid);
    if (ntt_tail_local<L>()) ntt_wave_sync(); else __syncthreads();       // pass C -> tail: block-local as well (tail_index)
    lds_get_tail<T, L>(x, s, tid);
    AR::template renorm_at<RS_FWD_TAIL>(x, m);
    fwd_tail<AR, L>(x, tw, m, tid);
}
// The same transform with a caller's hook between its passes (k_keyswitch_pair14: the NEXT digit's source words are requested at one hook
// and taken out of the memory pipeline at the next, so their latency is spent under a pass' arithmetic): hk.at<0>() in front of pass A,
// at<1..3>() behind the arithmetic of passes A, B, C.
template <class AR, int L, bool PRE, class HK> NTT_DEV void ntt_forward_regs_hooked(typename AR::T (&x)[16], typename AR::T *s, const typename AR::Tw &tw, const typename AR::Mod &m,
                                                                                     uint32_t tid, HK &hk) {
    typedef typename AR::T T;
    constexpr int SA = NttPlan<L>::SA;
    hk.template at<0>();
    fwd_stages<AR, L, SA, 0>(x, tw, m, tid);
    hk.template at<1>();
    if (PRE) __syncthreads();
    lds_put<T, L, SA, 0>(x, s, tid);
    __syncthreads();
    lds_get<T, L, 4, SA>(x, s, tid);
    AR::template renorm_at<RS_FWD_B>(x, m);
    fwd_stages<AR, L, 4, SA>(x, tw, m, tid);
    hk.template at<2>();
    lds_put<T, L, 4, SA>(x, s, tid);
    ntt_wave_sync();
    lds_get<T, L, 4, SA + 4>(x, s, tid);
    AR::template renorm_at<RS_FWD_C>(x, m);
    fwd_stages<AR, L, 4, SA + 4>(x, tw, m, tid);
    hk.template at<3>();
    lds_put<T, L, 4, SA + 4>(x, s, tid);
    if (ntt_tail_local<L>()) ntt_wave_sync(); else __syncthreads();
    lds_get_tail<T, L>(x, s, tid);
    AR::template renorm_at<RS_FWD_TAIL>(x, m);
    fwd_tail<AR, L>(x, tw, m, tid);
}
// Inverse transform (without the 1/N factor).  In: x[r] = value at position tail_index<L>(tid,r) (U64: [0,2q); F64: any
// |x| < 2^52).  Out: x[r] = coefficient pass_index<L,SA,0>(tid,r) (U64: [0,2q); F64: |x| <= 8.5q).
template <class AR, int L> NTT_DEV void ntt_inverse_regs(typename AR::T (&x)[16], typename AR::T *s, const typename AR::Tw &tw, const typename AR::Mod &m, uint32_t tid) {
    typedef typename AR::T T;
    constexpr int SA = NttPlan<L>::SA;
    AR::template renorm_at<RS_INV_START>(x, m);
    inv_tail<AR, L>(x, tw, m, tid);
    lds_put_tail<T, L>(x, s, tid);
    if (ntt_tail_local<L>()) ntt_wave_sync(); else __syncthreads();       // tail -> pass C: block-local (tail_index)
    lds_get<T, L, 4, SA + 4>(x, s, tid);
    AR::template renorm_at<RS_INV_C>(x, m);
    inv_stages<AR, L, 4, SA + 4>(x, tw, m, tid);
    lds_put<T, L, 4, SA + 4>(x, s, tid);
    ntt_wave_sync();                      // pass C -> B: wave-local, see ntt_forward_regs
    lds_get<T, L, 4, SA>(x, s, tid);
    AR::template renorm_at<RS_INV_B>(x, m);
    inv_stages<AR, L, 4, SA>(x, tw, m, tid);
    lds_put<T, L, 4, SA>(x, s, tid);
    __syncthreads();
    lds_get<T, L, SA, 0>(x, s, tid);
    AR::template renorm_at<RS_INV_A>(x, m);
    inv_stages<AR, L, SA, 0>(x, tw, m, tid);
}

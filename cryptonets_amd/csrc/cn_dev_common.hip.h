// Device-side helpers shared by every kernel family of libcnhip.so (gfx950 / CDNA4, wave64): modular arithmetic on u64 residues,
// per-policy views of the context constants, the counter-based sampler.  Everything here is inline / template code.
//
// Data layout in HBM: ciphertext array = [ct][poly][limb][N] u64 (SEAL's per-ciphertext layout,
// contiguous over the batch), so lane i of a wave touches coefficient i of one limb: every global
// access is a fully coalesced 8 B/lane (512 B/wave) stream.
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
DEV uint64_t canon4(uint64_t v, uint64_t q) { uint64_t q2 = 2 * q; v -= (v >= q2) ? q2 : 0; v -= (v >= q) ? q : 0; return v; }

// tw layout: modulus m -> tw + m*4n : w, ws, iw, iws
DEV const uint64_t *tw_of(const DevConsts *C, uint32_t mod) { return C->tw + (size_t)mod * 4 * C->n; }
struct Geo { uint32_t chunks, bs; };
DEV void decode(uint32_t chunks, uint32_t &limb, uint32_t &i) { limb = blockIdx.x / chunks; i = (blockIdx.x % chunks) * blockDim.x + threadIdx.x; }
DEV uint64_t scale_plain(const DevConsts *C, uint64_t m, uint32_t j) {       // Delta*m (+ r_t(q) in the upper half) mod q_j
    u128 p = (u128)C->delta[j] * m;
    if (m >= C->t_half) p += C->rtq[j];
    return bred128(p, C->q[j]);
}
// element-wise exact-FP64 modular arithmetic (per-coefficient kernels: GEMM fold, BEHZ extend / floor)
typedef ArF64T<1> BzF;
DEV double bz_canon(double x, const BzF::Mod &m) { double r = BzF::center(x, m); return r < 0.0 ? __dadd_rn(r, m.q) : r; }
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
// streams: 0 = ternary, 1/2 = noise polys, 3 = uniform.  `salt` (cn_set_rng_salt) whitens the counter word: with the 64-bit Philox key
// `seed` the sampler then depends on 128 secret bits (distinct (stream, trial) pairs stay distinct under the XOR)
DEV int32_t sample_ternary(uint64_t seed, uint64_t stream, uint64_t item, uint32_t i, uint64_t salt) {
    for (uint32_t tr = 0;; tr++) {
        Philox p = philox(seed, ((stream << 32) | tr) ^ salt, (item << 20) | i);
#pragma unroll
        for (int w = 0; w < 4; w++) for (int b = 0; b < 32; b += 2) { uint32_t v = (p.c[w] >> b) & 3; if (v != 3) return (int32_t)v - 1; }
    }
}
DEV int32_t sample_noise(uint64_t seed, uint64_t stream, uint64_t item, uint32_t i, uint64_t salt) {      // clipped normal sigma 3.2, 6 sigma, cast
    for (uint32_t tr = 0;; tr++) {
        Philox p = philox(seed, ((stream << 32) | tr) ^ salt, (item << 20) | i);
        const double u1 = ((double)(((uint64_t)p.c[0] << 21) ^ (p.c[1] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
        const double u2 = ((double)(((uint64_t)p.c[2] << 21) ^ (p.c[3] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
        const double g = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2) * 3.2;
        if (fabs(g) <= 19.2) return (int32_t)g;
    }
}
DEV uint64_t sample_uniform(uint64_t seed, uint64_t stream, uint64_t item, uint32_t i, uint64_t q, uint64_t salt) {
    const uint64_t lim = ~0ull - (~0ull % q) - 1;
    for (uint32_t tr = 0;; tr++) {
        Philox p = philox(seed, ((stream << 32) | tr) ^ salt, (item << 20) | i);
        uint64_t v = ((uint64_t)p.c[0] << 32) | p.c[1];
        if (v <= lim) return v % q;
        v = ((uint64_t)p.c[2] << 32) | p.c[3];
        if (v <= lim) return v % q;
    }
}

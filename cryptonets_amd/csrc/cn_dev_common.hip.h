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

// copy `words` doubles of a global table behind the exchange image (all threads; caller synchronises)
DEV void stage_table(double *dst, const NTT_GLOBAL double *src, uint32_t words, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid * 2; i < words; i += nthreads * 2) {
        const double a = src[i], b = src[i + 1];               // adjacent lanes, adjacent pairs: 16 B per lane either way
        dst[i] = a; dst[i + 1] = b;
    }
}
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
// KeyGenerator / Encryptor / Decryptor of the data owner, for deployments where the client has a GPU too.  Randomness: ChaCha20 (RFC 7539
// block function, 20 rounds) as a counter-mode DRBG - 256-bit key per context (cn_set_rng_key: the data owner draws it from the OS
// entropy source), 64-bit nonce per call (the `seed` argument of cn_keygen / cn_encrypt), 64-bit block counter made of
// (polynomial item, stream, redraw trial, block index inside the polynomial): every block of every polynomial of every call is distinct.
// One block (512 bits) yields 16 ternary coefficients (32 bits each: 16 two-bit rejection trials), 8 clipped-normal coefficients (four
// Box-Muller pairs, both branches used) or 8 uniform 64-bit words; secrets and noise are drawn ONCE per coefficient into int8 arrays
// and expanded to the k residues afterwards.
struct RngKey { uint32_t k[8]; };
// EncTab: per-ciphertext parameters of an encryption whose outputs are separate arrays and whose calls carried their own nonce (deferred
// per-ciphertext cn_encrypt calls merged into one launch chain): output address, plaintext polynomial or null, nonce, polynomial item
struct EncTab { NTT_GLOBAL uint64_t *out; const NTT_GLOBAL uint64_t *pt; uint64_t nonce, item; };
DEV uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
#define CN_QR(a, b, c, d) a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7);
DEV void chacha20_block(const RngKey &key, uint64_t counter, uint64_t nonce, uint32_t (&out)[16]) {
    uint32_t x[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3], key.k[4], key.k[5], key.k[6], key.k[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)nonce, (uint32_t)(nonce >> 32)};
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = x[i];
#pragma unroll 2
    for (int r = 0; r < 10; r++) {
        CN_QR(w[0], w[4], w[8], w[12]) CN_QR(w[1], w[5], w[9], w[13]) CN_QR(w[2], w[6], w[10], w[14]) CN_QR(w[3], w[7], w[11], w[15])
        CN_QR(w[0], w[5], w[10], w[15]) CN_QR(w[1], w[6], w[11], w[12]) CN_QR(w[2], w[7], w[8], w[13]) CN_QR(w[3], w[4], w[9], w[14])
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = w[i] + x[i];
}
// block counter: item (40 bits) | stream (4) | trial (4) | block index inside the polynomial (16)
DEV uint64_t rng_counter(uint64_t item, uint32_t stream, uint32_t trial, uint32_t blk) { return (item << 24) | ((uint64_t)(stream & 15) << 20) | ((uint64_t)(trial & 15) << 16) | (blk & 0xffffu); }
// streams: 0 = ternary (secret key, u), 1 / 2 = noise polynomials e1 / e2 (and key noise), 3 = uniform (the `a` component of keys)
// 16 ternary coefficients {-1, 0, 1} from one block: coefficient c takes word c, two bits at a time, the first pair that is not 3
DEV void sample_ternary16(const RngKey &key, uint64_t nonce, uint32_t stream, uint64_t item, uint32_t blk, int8_t (&out)[16]) {
    uint32_t w[16];
    chacha20_block(key, rng_counter(item, stream, 0, blk), nonce, w);
    uint32_t pending = 0;
#pragma unroll
    for (int c = 0; c < 16; c++) {
        int v = 2;                                                   // 2 = every pair was 3 (probability 4^-16): redraw below
        for (int b = 30; b >= 0; b -= 2) { const uint32_t t = (w[c] >> b) & 3; if (t != 3) v = (int)t - 1; }   // ends with the LOWEST non-3 pair
        out[c] = (int8_t)v;
        if (v == 2) pending |= 1u << c;
    }
    for (uint32_t trial = 1; pending; trial++) {                     // (practically never runs)
        chacha20_block(key, rng_counter(item, stream, trial, blk), nonce, w);
        for (int c = 0; c < 16; c++) if (pending & (1u << c)) {
            for (int b = 30; b >= 0; b -= 2) { const uint32_t t = (w[c] >> b) & 3; if (t != 3) { out[c] = (int8_t)((int)t - 1); pending &= ~(1u << c); } }
        }
    }
}
// 8 coefficients of the clipped normal (sigma 3.2, clipped at 6 sigma = 19.2, rounded towards zero like SEAL's static_cast of its ClippedNormalDistribution draw) from one block.
// Round 6: inversion of the cumulative distribution instead of Box-Muller in FP64 (log, sqrt, sinpi, cospi and a redraw loop per pair were ~200 FP64 instructions per coefficient - the
// noise sampler took 100 us per 1 290 polynomials, a fifth of an encryption).  One 64-bit word per coefficient: the top bit is the sign, the other 63 bits are compared with the 19
// thresholds thr[i] = floor(2^63 P(|x| < i + 1 | |x| <= 19.2)) (cn_noise_table(), long double erf): |value| = the number of thresholds at or below it.  The same distribution
// (P(0) = P(|x| < 1), P(+-k) = P(k <= |x| < k + 1) / 2, P(+-19) = P(19 <= |x| <= 19.2) / 2) to 2^-63, no rejection loop; a different stream than rounds 1-5 for the same seed.
struct NoiseTab { uint64_t thr[19]; };
DEV void sample_noise8(const RngKey &key, uint64_t nonce, uint32_t stream, uint64_t item, uint32_t blk, const NoiseTab &nt, int8_t (&out)[8]) {
    uint32_t w[16];
    chacha20_block(key, rng_counter(item, stream, 0, blk), nonce, w);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const uint64_t v = ((uint64_t)w[2 * c] << 32) | w[2 * c + 1], mag = v & 0x7fffffffffffffffull;
        int k = 0;
#pragma unroll
        for (int i = 0; i < 19; i++) k += nt.thr[i] <= mag ? 1 : 0;
        out[c] = (int8_t)((v >> 63) ? -k : k);
    }
}
// 8 uniform residues mod q from one block (64-bit words, rejection of the incomplete top range)
DEV void sample_uniform8(const RngKey &key, uint64_t nonce, uint32_t stream, uint64_t item, uint32_t blk, uint64_t q, uint64_t (&out)[8]) {
    const uint64_t lim = ~0ull - (~0ull % q) - 1;
    uint32_t pending = 0xff;
    for (uint32_t trial = 0; pending; trial++) {
        uint32_t w[16];
        chacha20_block(key, rng_counter(item, stream, trial, blk), nonce, w);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint64_t v = ((uint64_t)w[2 * c] << 32) | w[2 * c + 1];
            if ((pending & (1u << c)) && v <= lim) { out[c] = v % q; pending &= ~(1u << c); }
        }
    }
}

// Element-wise kernels, the radix-2 LDS transform (sizes without a register-radix kernel), Galois permutations, the legacy key switch
// and the client-side (keygen / decrypt) element-wise kernels.  Included (through cn_api_shared.h) by the runtime units; the kernels have internal linkage - every unit carries the ones it launches.
#pragma once
#include "cn_dev_common.hip.h"

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
// batched in-place NTT: block b transforms limb b; modulus = base_off + (b % nmod)
static __global__ void __launch_bounds__(1024) k_ntt(uint64_t *data, const DevConsts *__restrict__ C, uint32_t base_off, uint32_t nmod, int inverse) {
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
// op: 0 add, 1 sub, 2 negate(a).  a,b,out point at ciphertext arrays with `polys` polys each.
static __global__ void k_addsub(const uint64_t *a, const uint64_t *b, uint64_t *out, const DevConsts *__restrict__ C, uint32_t chunks, int op) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint64_t q = C->q[limb % C->k].q; size_t o = (size_t)limb * C->n + i;
    uint64_t x = a[o];
    out[o] = op == 0 ? addmod(x, b[o], q) : (op == 1 ? submod(x, b[o], q) : negmod(x, q));
}
// out = sum_i in[idx[i]] ; limbs = polys*k of ONE ciphertext
static __global__ void k_add_many(const uint64_t *in, const uint32_t *__restrict__ idx, uint32_t n_idx, size_t ct_words, uint64_t *out,
                           const DevConsts *__restrict__ C, uint32_t chunks) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint64_t q = C->q[limb % C->k].q; size_t o = (size_t)limb * C->n + i;
    uint64_t acc = 0;
    for (uint32_t t = 0; t < n_idx; t++) acc = addmod(acc, in[(size_t)idx[t] * ct_words + o], q);
    out[o] = acc;
}
// a: [count][polys][k][N]; pt: [..][N]; grid over count*polys*k limbs
static __global__ void k_add_plain(const uint64_t *a, const uint64_t *pt, uint32_t pt_stride_words, uint64_t *out, const DevConsts *__restrict__ C,
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
static __global__ void k_lift_plain(const uint64_t *pt, uint64_t *lifted, const DevConsts *__restrict__ C, uint32_t chunks, uint32_t pitch) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t k = C->k, j = limb % k, pi = limb / k;
    uint64_t m = pt[(size_t)pi * pitch * C->n + i];          // plaintext pi of the batch sits `pitch` plaintexts after plaintext pi-1
    lifted[(size_t)limb * C->n + i] = m >= C->t_half ? m + C->lift_inc[j] : m;
}
// x[ct][p][j][i] *= ptn[(ct*pstride)][j][i]   (both in NTT form)
static __global__ void k_dyadic_pt(uint64_t *x, const uint64_t *ptn, uint32_t pstride, const DevConsts *__restrict__ C, uint32_t chunks, uint32_t polys) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t k = C->k, j = limb % k, ct = limb / (k * polys);
    size_t o = (size_t)limb * C->n + i;
    x[o] = mulmod(x[o], ptn[((size_t)ct * pstride * k + j) * C->n + i], C->q[j]);
}
// out[ct] = a[ct] * lifted scalar sc[ct*sstride*k + j]   (constant-plaintext multiply_plain)
static __global__ void k_mul_scalar(const uint64_t *a, const uint64_t *__restrict__ sc, uint32_t sstride, uint64_t *out, const DevConsts *__restrict__ C,
                             uint32_t chunks, uint32_t polys) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t k = C->k, j = limb % k, ct = limb / (k * polys);
    size_t o = (size_t)limb * C->n + i;
    out[o] = mulmod(a[o], sc[(size_t)ct * sstride * k + j], C->q[j]);
}
// Step 2: tensor product in NTT form; A,B: [cnt][2][L][N], D: [cnt][3][L][N]; L limbs with moduli base_off..
static __global__ void k_tensor(const uint64_t *__restrict__ A, const uint64_t *__restrict__ B, uint64_t *__restrict__ D, const DevConsts *__restrict__ C,
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
// ------------------------------------------------------------------ key switching (relinearize / Galois)
// Fused per (ciphertext, output limb j): for every (source limb l, digit d): extract the base-2^dbc digit of
// target[l], NTT it under q_j in LDS, multiply-accumulate with the key pair K[(l,d)][0/1][j] (NTT form) in
// registers; finally INTT both accumulators and add them to add0/add1.  The digit polynomials never touch HBM.
// block = NT threads, EPT = N/NT accumulators per thread per output poly.
template <int EPT>
static __global__ void __launch_bounds__(1024) k_keyswitch(const uint64_t *__restrict__ target, size_t tgt_stride, const uint64_t *__restrict__ add0,
                                                    const uint64_t *__restrict__ add1, size_t add_stride, const uint64_t *__restrict__ key,
                                                    uint64_t *__restrict__ out, const DevConsts *__restrict__ C, int galois, uint64_t *const *__restrict__ out_tab) {
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
        const bool xi = C->ks_xi != 0;             // digits of [c_l (q/q_l)^-1]_{q_l} instead of c_l (cn_set_option("ks_xi"))
        const DMod ql = C->q[l]; const uint64_t xf = C->inv_qhat_q[l];
        for (uint32_t d = 0; d < nd; d++, kp += 2 * kn) {
            const int sh = dbc * (int)d;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                uint64_t v = src[tid + e * nt];
                if (xi) v = mulmod(v, xf, ql);
                v = (v >> sh) & mask;
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
        NTT_GLOBAL uint64_t *o = (NTT_GLOBAL uint64_t *)(out_tab ? out_tab[ct] : out + (size_t)ct * 2 * kn) + (size_t)p * kn + (size_t)j * n;
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
static __global__ void k_galois(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, const DevConsts *__restrict__ C, uint32_t chunks, uint64_t elt) {
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
static __global__ void __launch_bounds__(1024) k_galois_lds(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, const DevConsts *__restrict__ C, uint64_t elt) {
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
// The permutation of a rotation on SELECTED limbs (k_galois_lds, which permutes whole ciphertexts, with a stride): limb (ct, j) of
// src + ct*sstride -> dst + ct*dstride.  In front of k_keyswitch_pair14 only c1 is permuted ahead of time.
static __global__ void __launch_bounds__(1024) k_galois_limbs(const uint64_t *__restrict__ src, size_t sstride, uint64_t *__restrict__ dst, size_t dstride,
                                                       const DevConsts *__restrict__ C, uint64_t elt) {
    extern __shared__ uint64_t gsl[];
    const uint32_t n = C->n, k = C->k, ct = blockIdx.x / k, j = blockIdx.x % k, logn = C->logn;
    const uint64_t q = C->q[j].q;
    const uint64_t *x = src + (size_t)ct * sstride + (size_t)j * n;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t raw = (uint64_t)i * elt, v = x[i];
        gsl[(uint32_t)(raw & (n - 1))] = ((raw >> logn) & 1) ? negmod(v, q) : v;
    }
    __syncthreads();
    uint64_t *o = dst + (size_t)ct * dstride + (size_t)j * n;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) o[i] = gsl[i];
}
static __global__ void __launch_bounds__(256) k_ks_combine14(const uint64_t *__restrict__ half, const uint64_t *__restrict__ add0, const uint64_t *__restrict__ add1,
                                                       size_t add_stride, uint64_t *out, const DevConsts *__restrict__ C, const uint64_t *extra,
                                                       size_t ex_stride, uint64_t *const *__restrict__ out_tab) {
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
    NTT_GLOBAL uint64_t *o = (NTT_GLOBAL uint64_t *)(out_tab ? out_tab[ct] + ((size_t)p * k + j) * n : out + (size_t)limb * n);
    o[i] = lo; o[i + n2] = hi;
}
// in-place conversion of key words to the FP64 form used by k_keyswitch_rr<L, ArF64> (exact: residues < 2^49)
static __global__ void k_u64_to_f64(uint64_t *p, size_t words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) { double d = (double)(long long)p[i]; reinterpret_cast<double *>(p)[i] = d; }
}
// the way back (a key that arrived as an FP64 image in a context that runs the integer kernels: cn_ctx_broadcast_keys)
static __global__ void k_f64_to_u64(uint64_t *p, size_t words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) { double d = reinterpret_cast<double *>(p)[i]; p[i] = (uint64_t)(long long)d; }
}
// ---- samplers (ChaCha20 DRBG, cn_dev_common.hip.h).  Secrets and noise are drawn once per coefficient into int8 arrays.
// small[it][p][i], it < items, p < polys: kind 0 ternary {-1, 0, 1} (stream0 + p), kind 1 clipped normal (stream0 + p)
static __global__ void k_sample_small(int8_t *__restrict__ small, uint32_t n, int kind, uint32_t polys, uint32_t items, RngKey key, uint64_t nonce, uint32_t stream0,
                               uint64_t item0, const EncTab *__restrict__ tab, NoiseTab nt) {
    const uint32_t per = kind == 0 ? 16u : 8u, bpp = n / per;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)items * polys * bpp) return;
    const uint32_t blk = (uint32_t)(gid % bpp), pi = (uint32_t)(gid / bpp), it = pi / polys, p = pi % polys;
    const uint64_t item = tab ? tab[it].item : item0 + it, nc = tab ? tab[it].nonce : nonce;
    int8_t *o = small + (size_t)pi * n + (size_t)blk * per;
    if (kind == 0) {
        int8_t v[16];
        sample_ternary16(key, nc, stream0 + p, item, blk, v);
#pragma unroll
        for (int c = 0; c < 16; c++) o[c] = v[c];
    } else {
        int8_t v[8];
        sample_noise8(key, nc, stream0 + p, item, blk, nt, v);
#pragma unroll
        for (int c = 0; c < 8; c++) o[c] = v[c];
    }
}
// one raw generator block (known-answer self-test, cn_rng_selftest)
static __global__ void k_rng_block(RngKey key, uint64_t counter, uint64_t nonce, uint32_t *out) {
    uint32_t w[16];
    chacha20_block(key, counter, nonce, w);
    for (int i = 0; i < 16; i++) out[i] = w[i];
}
// out[it][j][i] = small[it][i] as a residue mod q_j
static __global__ void k_expand_small(const int8_t *__restrict__ small, uint64_t *__restrict__ out, const DevConsts *__restrict__ C, uint32_t chunks) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t k = C->k, j = limb % k; const uint64_t q = C->q[j].q;
    const int32_t s = small[(size_t)(limb / k) * C->n + i];
    out[(size_t)limb * C->n + i] = s >= 0 ? (uint64_t)s : q - (uint64_t)(-s);
}
// out[it][j][i]: uniform residues mod q_j (the `a` component of keys, directly in the NTT domain); thread = 8 coefficients of one limb
static __global__ void k_sample_uniform(uint64_t *__restrict__ out, const DevConsts *__restrict__ C, uint32_t items, RngKey key, uint64_t nonce, uint32_t stream, uint64_t item0) {
    const uint32_t n = C->n, k = C->k, bpl = n / 8;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)items * k * bpl) return;
    const uint32_t b = (uint32_t)(gid % bpl), limb = (uint32_t)(gid / bpl), j = limb % k;
    uint64_t v[8];
    sample_uniform8(key, nonce, stream, item0 + limb / k, j * bpl + b, C->q[j].q, v);
#pragma unroll
    for (int c = 0; c < 8; c++) out[(size_t)limb * n + (size_t)b * 8 + c] = v[c];
}
// b = -(a s + e) + f[limb] snew: the first component of a public key (f all zero) or of key-switch key (l, d) - f = 2^(dbc d) in limb l and zero
// elsewhere, or (q/q_l) 2^(dbc d) mod q_j under cn_set_option("ks_xi", 1) (again zero unless j == l: q_j divides q/q_l)
struct KeyFactors { uint64_t f[CN_MAXK]; };
static __global__ void k_key_b(const uint64_t *a, const uint64_t *e, const uint64_t *s, const uint64_t *snew, KeyFactors fac, uint64_t *b,
                        const DevConsts *__restrict__ C, uint32_t chunks) {
    uint32_t limb, i; decode(chunks, limb, i);
    const DMod qm = C->q[limb]; size_t o = (size_t)limb * C->n + i;
    uint64_t v = negmod(addmod(mulmod(a[o], s[o], qm), e[o], qm.q), qm.q);
    const uint64_t factor = fac.f[limb];
    if (factor) v = addmod(v, mulmod(snew[o], factor, qm), qm.q);
    b[o] = v;
}
static __global__ void k_mul_limbs(const uint64_t *a, const uint64_t *b, uint64_t *o, const DevConsts *__restrict__ C, uint32_t chunks) {   // NTT-form product, [k][N]
    uint32_t limb, i; decode(chunks, limb, i);
    size_t x = (size_t)limb * C->n + i; o[x] = mulmod(a[x], b[x], C->q[limb % C->k]);
}
// o[ct][j] = a[ct][j] * b[j] (+ add[ct][j]): b broadcast over ciphertexts, NTT form
static __global__ void k_mul_limbs_bcast(const uint64_t *a, const uint64_t *b, const uint64_t *add, uint64_t *o, const DevConsts *__restrict__ C, uint32_t chunks) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t j = limb % C->k; const DMod qm = C->q[j];
    size_t x = (size_t)limb * C->n + i;
    uint64_t v = mulmod(a[x], b[(size_t)j * C->n + i], qm);
    o[x] = add ? addmod(v, add[x], qm.q) : v;
}
// noise probe: acc[ct][j] <- t * (c0[ct][j] + acc[ct][j]) mod q_j  (the polynomial whose centred norm InvariantNoiseBudget measures)
static __global__ void k_noise_poly(const uint64_t *__restrict__ c0, size_t ct_stride, uint64_t *__restrict__ acc, const DevConsts *__restrict__ C, uint32_t chunks) {
    uint32_t limb, i; decode(chunks, limb, i);
    const uint32_t j = limb % C->k, ct = limb / C->k; const DMod qm = C->q[j];
    const size_t x = (size_t)limb * C->n + i;
    acc[x] = mulmod(addmod(c0[(size_t)ct * ct_stride + (size_t)j * C->n + i], acc[x], qm.q), C->t.q % qm.q, qm);
}
// decryption tail: x_j = c0_j + acc_j (coefficient form), then m = round(t*x/q) mod t by the {t, gamma} trick
template <int K>
static __global__ void __launch_bounds__(256) k_decrypt_scale(const uint64_t *__restrict__ c0, size_t ct_stride, const uint64_t *__restrict__ acc, uint64_t *__restrict__ plain,
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

// FP64 issue-rate probe (cn_fp64_issue_time): ILP independent chains of the 6-instruction exact modular multiply per thread, nothing else -
// the rate at which the FP64 pipe issues under the power state of the moment (the key switch's floor in bench.py is priced with it).
template <int ILP>
static __global__ void __launch_bounds__(512) k_fp64_probe(double *out, double w, double q, double qinv, int iters) {
    extern __shared__ double probe_lds[];               // (dynamic LDS only pins the occupancy: one workgroup per CU, two waves per SIMD)
    double a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) a[i] = (double)(threadIdx.x * 131 + i * 7 + 1);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            const double p = __dmul_rn(a[i], w);
            const double e = __fma_rn(a[i], w, -p);
            const double h = __builtin_rint(__dmul_rn(p, qinv));
            a[i] = __dadd_rn(__fma_rn(-h, q, p), e);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += a[i];
    if (s == 1.2345e-300) probe_lds[threadIdx.x] = s;   // never true: keeps the LDS allocation and the chains alive
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the same for full-rate 32-bit VALU instructions (add / xor / shift-add / and-or: what the key switch runs besides FP64 - digit extraction,
// addressing, selects): 4 interleaved chains x 6 instructions per iteration, written out so that the count is exact
static __global__ void __launch_bounds__(512) k_valu_probe(double *out, uint32_t c1, uint32_t c2, int iters) {
    extern __shared__ double probe_lds[];
    uint32_t a0 = threadIdx.x, a1 = threadIdx.x * 3 + 1, a2 = threadIdx.x * 5 + 2, a3 = threadIdx.x * 7 + 3;
    for (int it = 0; it < iters; it++)
        asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     "v_xor_b32 %0, %0, %5\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %5\n v_xor_b32 %3, %3, %5\n"
                     "v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4\n"
                     "v_and_or_b32 %0, %0, %5, %4\n v_and_or_b32 %1, %1, %5, %4\n v_and_or_b32 %2, %2, %5, %4\n v_and_or_b32 %3, %3, %5, %4\n"
                     "v_sub_u32 %0, %0, %5\n v_sub_u32 %1, %1, %5\n v_sub_u32 %2, %2, %5\n v_sub_u32 %3, %3, %5\n"
                     "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c1), "v"(c2));
    const double s = (double)(a0 ^ a1 ^ a2 ^ a3);
    if (s == 1.2345e-300) probe_lds[threadIdx.x] = s;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

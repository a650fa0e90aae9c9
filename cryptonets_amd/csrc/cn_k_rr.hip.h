// Register-radix transform kernels (N = 2^L, L = 10..14): batched NTT, fused dense MultiplyPlain, tensor product fused into the inverse
// transform, fused squaring, encryption tail.  Included by cn_l_rr.inc.h (one translation unit per arithmetic policy).
#pragma once
#include "cn_dev_common.hip.h"

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
// (Measured, not kept - round 2: the same transform on a RESIDENT grid, one 512-thread workgroup per CU walking limbs b, b + G, ... of one
// modulus with the twiddle table in LDS, the first-pass roots in SGPRs and the next limb's words prefetched a whole transform ahead.
// 311 us against 301 us for the 8450-limb batch: the time of a limb is not its HBM latency.  tools/ubench_stream.hip: the same access
// pattern with MORE FP64 work but no LDS exchanges runs at 212 us (5.2 TB/s), without arithmetic at 185 us (6.0 TB/s).)
// BEHZ step 2 fused into the inverse transform: block = (ciphertext, output poly p of the tensor product, limb).  The NTT-form
// operands are read at the positions the inverse transform starts from (16 B/lane), d_p = a0*b0 | a0*b1 + a1*b0 | a1*b1 is formed
// in registers and transformed back at once - the 3-poly NTT-form tensor never exists in HBM (saves one write + one read of
// 3(k + k+1) limbs per ciphertext and a kernel).  A, B: [cnt][2][Lm][N]; D: [cnt][3][Lm][N] (coefficient form, canonical).
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
// ONE ciphertext times MANY plaintexts (the row-dot batches of the LoLa dense layers, EncryptedSealBfvMatrix.cs:79-120 -> DotProduct: 5488 rows at CIFAR
// shapes) in one launch (round 5): the ciphertext's 2k limbs are transformed ONCE (`ctn`: NTT form, canonical, [polys][k][N] - 2 MiB that stay in L2),
// block = (row, poly, limb j) lifts the row's plaintext into q_j while loading it, transforms IT, multiplies by the ciphertext's NTT words at the positions
// the thread holds (16 B/lane), transforms back.  INTT(NTT(lift(pt)) . NTT(ct)): the same words as k_lift_ntt + k_mul_plain_fused, which transformed the
// broadcast ciphertext once per row and carried every lifted plaintext through HBM in NTT form (k limbs written + read twice per row: 5.5 GiB per call at CIFAR
// shapes) - a launch, a fifth of the transforms and that round trip less.
// next_elt != 0: the c1 limbs (poly 1) leave a second time, permuted by the Galois element of the rotation that follows (the first link of the row-dot batch's
// SumAllSlots chain, k_keyswitch_pair14): next_out[row][k][N] - staged through the exchange image (coalesced stores), no permutation pass in front of the chain.
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT) k_mul_plain_bcast(const uint64_t *__restrict__ pt, uint32_t pitch, const uint64_t *__restrict__ ctn, uint64_t *__restrict__ out,
                                                                    const DevConsts *__restrict__ C, uint32_t polys, uint32_t next_elt, uint64_t *__restrict__ next_out) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t tid = threadIdx.x, k = C->k, j = blockIdx.x % k, cp = blockIdx.x / k, row = cp / polys, p = cp % polys;
    const ArCtx<AR> A(C, j);
    const TensorOps<AR> ops(C, j);
    const uint64_t *x = pt + (size_t)row * pitch * n, th = C->t_half, inc = C->lift_inc[j];
    const uint64_t *w = ctn + ((size_t)p * k + j) * n;
    T v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { const uint64_t m = x[pass_index<L, SA, 0>(tid, r)]; v[r] = A.load(m >= th ? m + inc : m); }
    ntt_forward_regs<AR, L>(v, s, A.fw, A.m, tid);
    uint32_t tm = tid;
    asm volatile("" : "+v"(tm));
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const ulonglong2 y = *reinterpret_cast<const ulonglong2 *>(w + tail_index<L>(tm, r));
        if constexpr (std::is_same<T, double>::value) {                   // lazy transform output x canonical word: exact (see KsMac)
            v[r] = ops.mul(v[r], A.load(y.x), A); v[r + 1] = ops.mul(v[r + 1], A.load(y.y), A);
        } else {
            v[r] = ops.mul(A.canon(v[r]), y.x, A); v[r + 1] = ops.mul(A.canon(v[r + 1]), y.y, A);
        }
    }
    if (!ntt_tail_local<L>()) __syncthreads();
    uint32_t ti = tid;
    asm volatile("" : "+v"(ti));
    ntt_inverse_regs<AR, L>(v, s, A.iv, A.m, ti);
    uint64_t *o = out + ((size_t)cp * k + j) * n;
    const bool chain = next_elt && p == 1;
    uint64_t *win = reinterpret_cast<uint64_t *>(smem);                    // N words: inside the (padded) exchange image
    if (chain) __syncthreads();                                            // everybody has taken its coefficients out of the image
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t e = pass_index<L, SA, 0>(ti, r);
        const uint64_t val = A.scaled(v[r]);
        o[e] = val;
        if (chain) { const uint32_t pos = (e * next_elt) & (2 * n - 1); win[pos & (n - 1)] = (pos >> L) ? negmod(val, C->q[j].q) : val; }
    }
    if (chain) {
        __syncthreads();
        uint64_t *no = next_out + ((size_t)row * k + j) * n;
#pragma unroll
        for (int r = 0; r < 16; r++) no[ti + NttPlan<L>::NT * (uint32_t)r] = win[ti + NttPlan<L>::NT * (uint32_t)r];
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
// PLDS (N <= 8192): the NTT-form operand is parked in LDS behind the exchange image instead (the thread's own 16 B slots, no barrier):
// nothing but the algorithmic 2 reads + 3 writes per block reaches memory, at the price of ONE workgroup per CU (image + 8 N bytes =
// 132 KiB of the 160 KiB) instead of two.
template <int L, class AR, bool PLDS = false>
__global__ void __launch_bounds__(NttPlan<L>::NT, PLDS ? 2 : 4) k_square_fused(const uint64_t *__restrict__ A_, size_t a_stride, const uint64_t *const *__restrict__ a_tab,
                                                                    uint64_t *__restrict__ D, const DevConsts *__restrict__ C, uint32_t base_off, uint32_t Lm) {
    // 4 waves per SIMD: two 512-thread workgroups per CU.  a_stride: words between the operands of consecutive ciphertexts (the q side
    // reads the input ciphertexts in place, the Bsk side k_behz_extend's array); a_tab: one operand address per ciphertext instead
    // (deferred per-ciphertext calls: every input ciphertext is its own array)
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
    const NTT_GLOBAL uint64_t *a0 = (const NTT_GLOBAL uint64_t *)(a_tab ? a_tab[ct] : A_ + (size_t)ct * a_stride) + (size_t)l * n, *a1 = a0 + Ln;   // (global, not flat, loads)
    uint64_t *d0 = D + (size_t)ct * 3 * Ln + (size_t)l * n, *d1 = d0 + Ln, *d2 = d1 + Ln;
    struct alignas(16) P2 { T a, b; };
    P2 *pk = reinterpret_cast<P2 *>(s + ntt_lds_words(n)) + tid;                  // PLDS: slot (r >> 1) of this thread at pk[(r >> 1) * NT]
#pragma unroll 1
    for (int step = 0; step < 3; step++) {
        uint32_t tl = tid;
        asm volatile("" : "+v"(tl));                             // one transform's address math / twiddles live at a time
        T v[16];
        if (step < 2) {
            const NTT_GLOBAL uint64_t *x = step ? a1 : a0;
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = A.load(x[pass_index<L, SA, 0>(tl, r)]);
            ntt_forward_regs<AR, L, true>(v, s, A.fw, A.m, tl);  // PRE: the image of the previous inverse transform is free
            AR::renorm(v, A.m);                                  // lazy transform output (up to 28 q) -> |x| <= q/2
            if constexpr (PLDS) {
                if (step == 0) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) pk[(r >> 1) * NttPlan<L>::NT] = P2{v[r], v[r + 1]};
#pragma unroll
                    for (int r = 0; r < 16; r++) v[r] = AR::mulmod(v[r], v[r], A.m);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const P2 w = pk[(r >> 1) * NttPlan<L>::NT];                 // A0 out, A1 in: the same thread's slot
                        pk[(r >> 1) * NttPlan<L>::NT] = P2{v[r], v[r + 1]};
                        v[r] = AR::mulmod(__dadd_rn(w.a, w.a), v[r], A.m); v[r + 1] = AR::mulmod(__dadd_rn(w.b, w.b), v[r + 1], A.m);
                    }
                }
            } else {
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
            }
            if (!ntt_tail_local<L>()) __syncthreads();           // (block-local tail: the inverse starts inside the wave's own blocks)
        } else {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                T wa, wb;
                if constexpr (PLDS) { const P2 w = pk[(r >> 1) * NttPlan<L>::NT]; wa = w.a; wb = w.b; }
                else {
                    const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(d2 + tail_index<L>(tl, r));
                    wa = __longlong_as_double((long long)w.x); wb = __longlong_as_double((long long)w.y);
                }
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
// The same squaring as a PIPELINED, RESIDENT kernel (round 4; N <= 8192, FP64 policies).  k_square_fused<.., PLDS> runs at 0.5 of its FP64 issue
// floor with one workgroup per CU: every step loads its operand and waits, every pass of every transform fetches its roots from L2 and waits, with two
// waves per SIMD to hide either (profiles/r03_bench_kernel_trace_summary.txt: 736 / 900 us for the q / Bsk side of the 845-ciphertext layer against an
// issue floor of 349 / 419 us).  Here:
//   * the grid is RESIDENT: gridDim.x = Lm * G workgroups (one per CU), workgroup (l, w) squares limb l of ciphertexts w, w + G, ... - one modulus per
//     workgroup for its whole life, so
//   * the INVERSE root table of that modulus is copied into LDS once (8 N bytes behind the exchange image - the room the parked operand took) and the
//     three inverse transforms of every block read their roots with ds_read instead of global loads; the first-pass roots of the forward transforms
//     live in SGPRs (ArPassA) - the two devices of the fused key switch (cn_k_ks.hip.h);
//   * the NTT-form operand is parked in REGISTERS (16 doubles: one workgroup per CU leaves 256 VGPRs per thread);
//   * the NEXT operand (a1 of this block, a0 of the workgroup's next block) is requested right behind the forward transform - after its last root load,
//     so that no later wait on a vector-memory counter has to drain it early (the counter is in-order) - and arrives under the tensor, the inverse
//     transform (no vector-memory loads any more) and the stores.
// Same words as the separate launches (tests/test_gpu_evaluator.py::test_squaring_fused_kernel_is_the_separate_launches).
template <class AR> struct LdsTwiddles;
template <int RN> struct LdsTwiddles<ArF64T<RN>> { typedef ArF64LdsT<RN> type; };
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT, 2) k_square_pipe(const uint64_t *__restrict__ A_, size_t a_stride, const uint64_t *const *__restrict__ a_tab,
                                                                    uint64_t *__restrict__ D, const DevConsts *__restrict__ C, uint32_t base_off, uint32_t Lm, uint32_t cnt) {
    typedef typename AR::T T;
    static_assert(std::is_same<T, double>::value && L <= 13, "FP64 policies, N <= 8192");
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t tid = threadIdx.x;
    const uint32_t G = gridDim.x / Lm, l = blockIdx.x % Lm, w0 = blockIdx.x / Lm, mod = base_off + l;
    const ArCtx<AR> A(C, mod);
    const size_t Ln = (size_t)Lm * n;
    typedef ArPassA<AR> FW;
    typename FW::Tw fwt;
    static_cast<typename AR::Tw &>(fwt) = A.fw;
    ntt_load_pass_a<SA>(fwt, A.fw.w);
    typedef typename LdsTwiddles<AR>::type IV;
    typename IV::Tw ivt;
    {
        double *tws = reinterpret_cast<double *>(smem) + ntt_lds_words(n);
        stage_table(tws, A.iv.w, n, tid, NttPlan<L>::NT);
        ivt.w = (const __attribute__((address_space(3))) double *)tws;
        __syncthreads();
    }
    auto operand = [&](uint32_t ct) { return (const NTT_GLOBAL uint64_t *)(a_tab ? a_tab[ct] : A_ + (size_t)ct * a_stride) + (size_t)l * n; };
    uint64_t nxt[16];
    {                                                            // (the launcher guarantees G <= cnt: every workgroup has a first block)
        const NTT_GLOBAL uint64_t *x = operand(w0);
#pragma unroll
        for (int r = 0; r < 16; r++) nxt[r] = x[pass_index<L, SA, 0>(tid, r)];
    }
    for (uint32_t ct = w0; ct < cnt; ct += G) {
        uint64_t *d0 = D + (size_t)ct * 3 * Ln + (size_t)l * n;
        T P[16];                                                 // the parked NTT-form operand: A0 after step 0, A1 after step 1
#pragma unroll 1
        for (int step = 0; step < 3; step++) {
            uint32_t tl = tid;
            asm volatile("" : "+v"(tl));                         // one transform's address math / twiddles live at a time
            T v[16];
            if (step < 2) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = A.load(nxt[r]);
                ntt_forward_regs<FW, L, true>(v, s, fwt, A.m, tl);   // PRE: the image of the previous inverse transform is free
                AR::renorm(v, A.m);
                // the next operand: a1 of this ciphertext, or a0 of the workgroup's next one.  UNCONDITIONAL (the last block of a workgroup re-reads its
                // own a0 and drops it): behind a branch the wait-count pass has to assume the path without these loads and makes every later wait
                // drain them - and the stores behind them - early
                {
                    const uint32_t nct = step ? (ct + G < cnt ? ct + G : ct) : ct;
                    const NTT_GLOBAL uint64_t *x = operand(nct) + (step ? 0 : Ln);
                    uint32_t tp = tid;
                    asm volatile("" : "+v"(tp));
#pragma unroll
                    for (int r = 0; r < 16; r++) nxt[r] = x[pass_index<L, SA, 0>(tp, r)];
                }
                if (step == 0) {
#pragma unroll
                    for (int r = 0; r < 16; r++) { P[r] = v[r]; v[r] = AR::mulmod(v[r], v[r], A.m); }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r++) { const T a0 = P[r]; P[r] = v[r]; v[r] = AR::mulmod(__dadd_rn(a0, a0), v[r], A.m); }
                }
                if (!ntt_tail_local<L>()) __syncthreads();
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = AR::mulmod(P[r], P[r], A.m);
                __syncthreads();                                 // no forward transform in this step: the previous inverse's image is free
            }
            uint32_t ti = tid;
            asm volatile("" : "+v"(ti));
            ntt_inverse_regs<IV, L>(v, s, ivt, A.m, ti);
            uint64_t *o = d0 + (size_t)step * Ln;
            uint32_t to = tid;
            asm volatile("" : "+v"(to));
#pragma unroll
            for (int r = 0; r < 16; r++) o[pass_index<L, SA, 0>(to, r)] = A.scaled(v[r]);
        }
    }
}
// One output word of an encryption: v N^-1 + e (+ Delta m + r_t(q) in the upper half) as a canonical residue.  FP64 policies (round 6): the whole sum in exact doubles and ONE
// canonicalisation - Delta_j m is an error-free FP64 product (m < t < q_j < 2^49) instead of a 64 x 64 -> 128-bit multiply and a Barrett reduction per coefficient (~12 quarter-rate
// v_mad_u64_u32: with a dense plaintext the integer epilogue was a fourth transform's worth of issue slots - 659 us per 784 ciphertexts against 447 at the rate of the zero vectors).
template <class AR> struct EncTail {
    typedef typename AR::T T;
    const ArCtx<AR> &A; const DevConsts *C; uint32_t j; uint64_t q, t_half; double dlt, rtq;
    NTT_DEV EncTail(const ArCtx<AR> &A_, const DevConsts *C_, uint32_t j_, uint64_t q_) : A(A_), C(C_), j(j_), q(q_), t_half(C_->t_half), dlt((double)C_->delta[j_]), rtq((double)C_->rtq[j_]) {}
    NTT_DEV uint64_t word(T v, int32_t ns, bool has_m, uint64_t mw) const {
        if constexpr (std::is_same<T, double>::value) {
            double x = __dadd_rn(AR::mulmod(v, A.ni, A.m), (double)ns);
            if (has_m) {
                double y = AR::mulmod(AR::from_u64(mw), dlt, A.m);
                if (mw >= t_half) y = __dadd_rn(y, rtq);
                x = __dadd_rn(x, y);
            }
            return AR::to_u64(x, A.m);
        } else {
            uint64_t val = A.scaled(v);
            val = addmod(val, ns >= 0 ? (uint64_t)ns : q - (uint64_t)(-ns), q);
            if (has_m) val = addmod(val, scale_plain(C, mw, j), q);
            return val;
        }
    }
};
// encryption tail: out[ct][p][j] = INTT(u[ct][j] * pk[p][j]) + e_p (+ Delta*m for p = 0); u in NTT form, e = the int8 noise polynomials
// [ct][2][N] of k_sample_small.  tab: per-ciphertext output address and plaintext (deferred per-ciphertext calls), else out + ct*2kN and
// pt + ct*pt_stride_words (pt null: encryptions of zero)
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT) k_encrypt_tail(const uint64_t *__restrict__ u, const uint64_t *__restrict__ pk, const uint64_t *__restrict__ pt,
                                                                 uint32_t pt_stride_words, uint64_t *__restrict__ out, const DevConsts *__restrict__ C,
                                                                 const int8_t *__restrict__ noise, const EncTab *__restrict__ tab) {
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
    NTT_GLOBAL uint64_t *o = (tab ? tab[ct].out : (NTT_GLOBAL uint64_t *)out + (size_t)ct * 2 * k * n) + ((size_t)p * k + j) * n;
    const NTT_GLOBAL uint64_t *m = tab ? tab[ct].pt : (pt ? (const NTT_GLOBAL uint64_t *)pt + (size_t)ct * pt_stride_words : nullptr);
    const int8_t *ee = noise + ((size_t)ct * 2 + p) * n;
    const uint64_t q = C->q[j].q;
    const EncTail<AR> tail(A, C, j, q);
    const bool has_m = p == 0 && m;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t e = pass_index<L, SA, 0>(tid, r);
        o[e] = tail.word(v[r], ee[e], has_m, has_m ? m[e] : 0);
    }
}

// Encryptor.Encrypt in ONE kernel behind the samplers (round 4): block = (ciphertext, limb j).  The ternary u arrives as the sampler's int8 polynomial, becomes
// residues mod q_j in registers, is transformed ONCE and stays in registers (16 values) for both components: out[ct][p][j] = INTT(NTT(u) * pk[p][j]) + e_p
// (+ Delta m for p = 0).  Replaces k_expand_small (u as k limbs of u64 in HBM), the batched forward transform over those limbs and k_encrypt_tail (which read
// them back twice): 1 forward + 2 inverse transforms per block like before, none of the 3 x k limb round trips of u, 3 launches fewer.  The unchanged PoolLayer
// encrypts a zero vector per padded convolution tap (PoolLayer.cs:67-80: 1 290 per CryptoNets batch) - this chain was ~1 ms per plaintext prime and batch of
// its extra time (profiles/r04_unchanged_caller_replay.txt).  Same words as the three-launch chain (exact arithmetic; tests/test_gpu_client.py).
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT) k_encrypt_fused(const int8_t *__restrict__ us, const uint64_t *__restrict__ pk, const uint64_t *__restrict__ pt,
                                                                  uint32_t pt_stride_words, uint64_t *__restrict__ out, const DevConsts *__restrict__ C,
                                                                  const int8_t *__restrict__ noise, const EncTab *__restrict__ tab) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t k = C->k, tid = threadIdx.x, j = blockIdx.x % k, ct = blockIdx.x / k;
    const ArCtx<AR> A(C, j);
    const TensorOps<AR> ops(C, j);
    const uint64_t q = C->q[j].q;
    T U[16];
    {
        const int8_t *uu = us + (size_t)ct * n;
#pragma unroll
        for (int r = 0; r < 16; r++) { const int32_t v = uu[pass_index<L, SA, 0>(tid, r)]; U[r] = A.load(v >= 0 ? (uint64_t)v : q - (uint64_t)(-v)); }
        ntt_forward_regs<AR, L>(U, s, A.fw, A.m, tid);
        if constexpr (std::is_same<T, double>::value) AR::renorm(U, A.m);      // lazy transform output -> |x| <= q/2: multiplied twice below
        else {
#pragma unroll
            for (int r = 0; r < 16; r++) U[r] = A.canon(U[r]);
        }
    }
    NTT_GLOBAL uint64_t *obase = tab ? tab[ct].out : (NTT_GLOBAL uint64_t *)out + (size_t)ct * 2 * k * n;
    const NTT_GLOBAL uint64_t *m = tab ? tab[ct].pt : (pt ? (const NTT_GLOBAL uint64_t *)pt + (size_t)ct * pt_stride_words : nullptr);
#pragma unroll 1
    for (int p = 0; p < 2; p++) {
        uint32_t tl = tid;
        asm volatile("" : "+v"(tl));                                           // one transform's address math live at a time
        const uint64_t *pp = pk + ((size_t)p * k + j) * n;
        T v[16];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const ulonglong2 y = *reinterpret_cast<const ulonglong2 *>(pp + tail_index<L>(tl, r));
            v[r] = ops.mul(U[r], A.load(y.x), A); v[r + 1] = ops.mul(U[r + 1], A.load(y.y), A);
        }
        if (p || !ntt_tail_local<L>()) __syncthreads();                        // the image of the previous transform is free (block-local tail: the first inverse starts inside the wave's own blocks)
        ntt_inverse_regs<AR, L>(v, s, A.iv, A.m, tl);
        NTT_GLOBAL uint64_t *o = obase + ((size_t)p * k + j) * n;
        const int8_t *ee = noise + ((size_t)ct * 2 + p) * n;
        const EncTail<AR> tail(A, C, j, q);
        const bool has_m = p == 0 && m;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t e = pass_index<L, SA, 0>(tl, r);
            o[e] = tail.word(v[r], ee[e], has_m, has_m ? m[e] : 0);
        }
    }
}

// The same encryption with a block per (ciphertext, COMPONENT, limb) (round 6; cn_set_option("enc_fused", 2), the default of the FP64 policies: 784 dense plaintexts
// 420 us against 542 for k_encrypt_fused and 534 for the three launches, 645 zero vectors 311 / 372 / 401 - profiles/r06_encrypt_probe.txt): every block transforms u itself (4 transforms
// per (ciphertext, limb) instead of 3) but nothing has to survive an inverse transform - ~100 instead of 198 VGPRs, so TWO workgroups share a CU and each other's
// memory round trips and barriers.  Same words.
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT, 4) k_encrypt_split(const int8_t *__restrict__ us, const uint64_t *__restrict__ pk, const uint64_t *__restrict__ pt,
                                                                     uint32_t pt_stride_words, uint64_t *__restrict__ out, const DevConsts *__restrict__ C,
                                                                     const int8_t *__restrict__ noise, const EncTab *__restrict__ tab) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t k = C->k, tid = threadIdx.x, j = blockIdx.x % k, p = (blockIdx.x / k) & 1, ct = blockIdx.x / (2 * k);
    const ArCtx<AR> A(C, j);
    const TensorOps<AR> ops(C, j);
    const uint64_t q = C->q[j].q;
    T v[16];
    {
        const int8_t *uu = us + (size_t)ct * n;
#pragma unroll
        for (int r = 0; r < 16; r++) { const int32_t x = uu[pass_index<L, SA, 0>(tid, r)]; v[r] = A.load(x >= 0 ? (uint64_t)x : q - (uint64_t)(-x)); }
        ntt_forward_regs<AR, L>(v, s, A.fw, A.m, tid);
        if constexpr (std::is_same<T, double>::value) AR::renorm(v, A.m);
        else {
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = A.canon(v[r]);
        }
    }
    const uint64_t *pp = pk + ((size_t)p * k + j) * n;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const ulonglong2 y = *reinterpret_cast<const ulonglong2 *>(pp + tail_index<L>(tid, r));
        v[r] = ops.mul(v[r], A.load(y.x), A); v[r + 1] = ops.mul(v[r + 1], A.load(y.y), A);
    }
    if (!ntt_tail_local<L>()) __syncthreads();
    ntt_inverse_regs<AR, L>(v, s, A.iv, A.m, tid);
    NTT_GLOBAL uint64_t *o = (tab ? tab[ct].out : (NTT_GLOBAL uint64_t *)out + (size_t)ct * 2 * k * n) + ((size_t)p * k + j) * n;
    const NTT_GLOBAL uint64_t *m = tab ? tab[ct].pt : (pt ? (const NTT_GLOBAL uint64_t *)pt + (size_t)ct * pt_stride_words : nullptr);
    const int8_t *ee = noise + ((size_t)ct * 2 + p) * n;
    const EncTail<AR> tail(A, C, j, q);
    const bool has_m = p == 0 && m;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t e = pass_index<L, SA, 0>(tid, r);
        o[e] = tail.word(v[r], ee[e], has_m, has_m ? m[e] : 0);
    }
}

// Weighted sums of FRESH ENCRYPTIONS OF ZERO that nobody else ever reads, folded by linearity (round 6; FP64 policies).  The unchanged PoolLayer makes a fresh
// Encrypt(0) per padded convolution tap (PoolLayer.cs:67-80), multiplies it by the tap's weight inside DenseMatrixBySparseVectorMultiply and disposes of it after the
// layer: 645 encryptions per plaintext prime and batch that exist only as terms of 125 scalar products.  With E_t = (pk0 u_t + e1_t, pk1 u_t + e2_t):
//     sum_t w_t E_t = (pk0 (sum_t w_t u_t) + sum_t w_t e1_t,  pk1 (sum_t w_t u_t) + sum_t w_t e2_t)   mod q_j, EXACTLY
// - every operation is exact modular arithmetic and the results are canonical residues, so the output words are those of the literal evaluation (same sampler
// draws: every folded encryption keeps its nonce and item) while one forward and two inverse transforms serve all padded taps of an output instead of three per
// tap, and the scalar product no longer reads 645 extra ciphertexts.  Block = (folded output, limb j): U = sum_t w_t u_t from the samplers' int8 polynomials
// (|w_t| <= t/2 as a centred double, |u| <= 1, |e| <= 19: the host checks terms * t/2 * 20 < 2^52), recentred, transformed once, both components added ONTO the
// output the scalar product (and its folded bias) wrote in the launch before.  tests/test_deferred.py: words identical with the fold switched off.
// (FoldOut / FoldTerm: cn_runtime.h)
template <int L, class AR>
__global__ void __launch_bounds__(NttPlan<L>::NT, 4) k_encrypt_fold(const int8_t *__restrict__ us, const uint64_t *__restrict__ pk, const DevConsts *__restrict__ C,
                                                                    const int8_t *__restrict__ noise, const FoldOut *__restrict__ fout, const FoldTerm *__restrict__ terms) {
    // block = (folded output, component p, limb j) since the closing visit of round 6: like k_encrypt_split every block forms and transforms U itself, nothing survives the inverse
    // transform, 128 VGPRs, two workgroups per CU (one block for both components: 232 VGPRs, 468 us per 125 outputs beside the other prime's kernels)
    typedef typename AR::T T;
    static_assert(std::is_same<T, double>::value, "FP64 policies only");
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t k = C->k, tid = threadIdx.x, j = blockIdx.x % k, p = (blockIdx.x / k) & 1, ct = blockIdx.x / (2 * k);
    const ArCtx<AR> A(C, j);
    const TensorOps<AR> ops(C, j);
    const FoldOut fo = fout[ct];
    const FoldTerm *tt = terms + fo.first;
    // (three terms' byte loads in flight at a time: a term is 16 dependent-free one-byte loads and one round trip; a border output of the 5x5 stride-2 convolution has 5 or 9)
    auto accumulate = [&](T (&X)[16], const int8_t *base, size_t pitch, size_t off) {
        uint32_t t = 0;
#pragma unroll 1
        for (; t + 3 <= fo.count; t += 3) {
            int8_t b[3][16];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int8_t *pp = base + (size_t)tt[t + c].enc * pitch + off;
#pragma unroll
                for (int r = 0; r < 16; r++) b[c][r] = pp[pass_index<L, SA, 0>(tid, r)];
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double w = tt[t + c].w;
#pragma unroll
                for (int r = 0; r < 16; r++) X[r] = __fma_rn(w, (double)(int32_t)b[c][r], X[r]);
            }
        }
#pragma unroll 1
        for (; t < fo.count; t++) {
            const double w = tt[t].w;
            const int8_t *pp = base + (size_t)tt[t].enc * pitch + off;
#pragma unroll
            for (int r = 0; r < 16; r++) X[r] = __fma_rn(w, (double)(int32_t)pp[pass_index<L, SA, 0>(tid, r)], X[r]);
        }
    };
    T v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = 0.0;
    accumulate(v, us, n, 0);
    AR::renorm(v, A.m);                                                       // |U| <= q/2: a transform input
    ntt_forward_regs<AR, L>(v, s, A.fw, A.m, tid);
    AR::renorm(v, A.m);
    const uint64_t *pp = pk + ((size_t)p * k + j) * n;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const ulonglong2 y = *reinterpret_cast<const ulonglong2 *>(pp + tail_index<L>(tid, r));
        v[r] = ops.mul(v[r], A.load(y.x), A); v[r + 1] = ops.mul(v[r + 1], A.load(y.y), A);
    }
    if (!ntt_tail_local<L>()) __syncthreads();
    ntt_inverse_regs<AR, L>(v, s, A.iv, A.m, tid);
    T E[16];
#pragma unroll
    for (int r = 0; r < 16; r++) E[r] = 0.0;
    accumulate(E, noise, (size_t)2 * n, (size_t)p * n);
    NTT_GLOBAL uint64_t *o = (NTT_GLOBAL uint64_t *)fo.out + ((size_t)p * k + j) * n;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t e = pass_index<L, SA, 0>(tid, r);
        // v N^-1 + E + the word the scalar product wrote: exact doubles (|E| < 2^47, the word < q), one canonicalisation
        const double x = __dadd_rn(__dadd_rn(AR::mulmod(v[r], A.ni, A.m), AR::center(E[r], A.m)), AR::from_u64(o[e]));
        o[e] = AR::to_u64(x, A.m);
    }
}

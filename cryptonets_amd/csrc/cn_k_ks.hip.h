// HOT LOOP B/C: key switching (relinearisation, Galois) on the register-radix core.  Included by cn_l_ks.inc.h (one translation unit
// per arithmetic policy).
#pragma once
#include "cn_dev_common.hip.h"

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
// cn_set_option("ks_xi", 1): the digits of a source limb are those of xi_l = [c_l (q/q_l)^-1]_{q_l} instead of those of c_l (DevConsts::ks_xi) - one
// exact modular product per source word, once per (workgroup, source limb); wave-uniform branch, off by default
DEV void ks_premultiply(uint64_t (&raw)[16], const DevConsts *C, uint32_t l) {
    if (C->ks_xi) {
        const DMod ql = C->q[l]; const uint64_t f = C->inv_qhat_q[l];
#pragma unroll
        for (int r = 0; r < 16; r++) raw[r] = mulmod(raw[r], f, ql);
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
#ifndef KS_SRC_AHEAD
#define KS_SRC_AHEAD 0      // k_keyswitch_rr: the source words of the next limb requested a limb ahead (250 instead of 212 VGPRs, no spills).  Measured in round 5
                            // (tools/gpu_r05_aa.sh, alternating runs on one box) and NOT the default: 3.29-3.33 against 3.25-3.29 ms per 845-ciphertext launch - the
                            // round trip at the start of a limb is not what the kernel waits for
#endif
#ifndef KS_MAC_FENCE
#define KS_MAC_FENCE 0      // FP64 path: letting the scheduler interleave key loads with the MACs measured 11-14 % faster (same VGPRs)
#endif
// (Measured, not kept - round 2: the first key component of a digit requested BEFORE the digit's transform into registers nobody else
// uses, read back behind it.  Bit-exact; 3.37 vs 3.33 ms: the key stream costs bandwidth on the vector memory path, not exposed latency.)
// XI: the "ks_xi" decomposition (digits of [c_l (q/q_l)^-1]_{q_l}) as a separate instantiation - the default kernel is instruction for instruction the one
// that is profiled and priced (tools/ks_isa_counts.py)
template <int L, class AR, int MINW = 1, bool TWL = false, bool XI = false>
__global__ void __launch_bounds__(NttPlan<L>::NT, MINW) k_keyswitch_rr(const uint64_t *__restrict__ target, size_t tgt_stride, const uint64_t *__restrict__ add0,
                                                                 const uint64_t *__restrict__ add1, size_t add_stride, const void *__restrict__ key_,
                                                                 uint64_t *out, const DevConsts *__restrict__ C, int galois, uint32_t accmax,
                                                                 const uint64_t *extra, size_t ex_stride, uint64_t *const *__restrict__ out_tab,
                                                                 uint32_t xcd_cts) {
    typedef typename AR::T T;
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t k = C->k, tid = threadIdx.x;
    // out_tab: one output address per ciphertext instead of out + ct*2kN (deferred per-ciphertext calls)
    // (ct, j) with j fastest: the k workgroups of a ciphertext run together and share its source limbs through L2 / MALL.  (A
    // limb-major order that lets an XCD's workgroups share one key slice in L2 was measured: no gain at N = 8192 - the 15.6 MB of
    // keys stream from the infinity cache fast enough - and 30 % slower at N = 16384, where the source limbs then come from HBM
    // once per (limb, half) workgroup.)
    // xcd_cts (a multiple of 8, 0 = off): XCD-aware placement for the first xcd_cts ciphertexts - block b runs on XCD b % 8 (observed
    // dispatch rule; a speed assumption only), so the k workgroups of ONE ciphertext are given block ids of one residue class: they
    // follow each other on one XCD and the k - 1 later ones find the ciphertext's source limbs in THAT XCD's L2 instead of fetching
    // them through k different L2s.  Bijective: b -> (x = b % 8, s = b / 8) -> ct = 8 (s / k) + x, j = s % k.
    // xcd_cts = 0x80000000: limb-major order (all ciphertexts of output limb 0, then limb 1, ...): every XCD works on ONE key slice
    // (3.2 MiB at N = 8192, k = 5) at a time, which then stays in its 4 MiB L2; the source limbs are fetched once per output limb instead.
    uint32_t ct = blockIdx.x / k, j = blockIdx.x % k;
    if (xcd_cts & 0x80000000u) { const uint32_t cnt = gridDim.x / k; j = blockIdx.x / cnt; ct = blockIdx.x % cnt; }
    else if (blockIdx.x < xcd_cts * k) { const uint32_t x = blockIdx.x & 7, sl = blockIdx.x >> 3; ct = 8 * (sl / k) + x; j = sl % k; }
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
    uint64_t raw[16];                              // the source words of limb l stay in registers for all of its digits
    // AHEAD (KS_SRC_AHEAD, off): the source words of limb l + 1 requested when limb l starts.  Unconditional (the last limb requests itself again): a request
    // under an `if` would make the compiler's wait counts at the join assume the path without it.
    constexpr bool AHEAD = KS_SRC_AHEAD && MINW == 1 && L <= 13;
    uint64_t nraw[AHEAD ? 16 : 1];
    auto request = [&](uint64_t (&w)[16], uint32_t l) {
        uint32_t t0 = tid;
        asm volatile("" : "+v"(t0));
        const uint64_t *src = target + (size_t)ct * tgt_stride + (size_t)l * n;
#pragma unroll
        for (int r = 0; r < 16; r++) w[r] = src[pass_index<L, SA, 0>(t0, r)];
    };
    if constexpr (AHEAD) request(nraw, 0);
    for (uint32_t l = 0; l < k; l++) {
        const uint32_t nd = galois ? C->gk_dig[l] : C->rl_dig[l];
        if constexpr (AHEAD) {
#pragma unroll
            for (int r = 0; r < 16; r++) raw[r] = nraw[r];
            request(nraw, min(l + 1, k - 1));
        } else request(raw, l);
        if constexpr (XI) {
            const DMod ql = C->q[l]; const uint64_t xf = C->inv_qhat_q[l];
#pragma unroll
            for (int r = 0; r < 16; r++) raw[r] = mulmod(raw[r], xf, ql);
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
        NTT_GLOBAL uint64_t *o = (NTT_GLOBAL uint64_t *)(out_tab ? out_tab[ct] : out + (size_t)ct * 2 * kn) + (size_t)p * kn + (size_t)j * n;
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
// XI: the "ks_xi" convention as a second instantiation (the premultiplication sits inside the digit loop here - as a run-time branch it cost the default
// kernel 30-70 VGPRs)
template <class AR, bool XI = false>
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
                uint64_t sx = src[e], sy = src[e + n2];
                if constexpr (XI) { const DMod ql = C->q[l]; const uint64_t xf = C->inv_qhat_q[l]; sx = mulmod(sx, xf, ql); sy = mulmod(sy, xf, ql); }   // digits of xi_l
                T X = A.load((sx >> sh) & mask), Y = A.load((sy >> sh) & mask);
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
// N = 16384 key switch as ONE launch (round 5).  block = (ct, output limb j) runs BOTH 8192-point halves one after the other on the
// machinery of k_keyswitch_split14 (512 threads, one accumulator pair live at a time, no scratch): half 0's two inverse sub-transforms
// are parked (recentred doubles, `stash`, written and read back by the same thread), half 1's meet them in registers, and the last
// inverse stage (u + v, (u - v) w^-1), the N^-1 scaling and the addends are applied on the way out - what k_ks_combine14 did in a
// second pass over 2 x 2 MiB per ciphertext.  Rotations: `target` is sigma(c1) (permuted ONCE per ciphertext - by k_galois_limbs in front
// of a single rotation, or by the previous link of a rotate-and-add chain, see next_elt), add0 is the UNPERMUTED c0 and perm_elt the
// element: the workgroup that owns output limb j stages c0's limb j through LDS at its permuted positions (coalesced reads, odd-stride
// conflict-free LDS writes) - the only part of k_galois_lds that is not redundant across the k output limbs.  next_elt != 0: the new
// c1 limb leaves a second time, permuted by the NEXT rotation's element, into next_out[ct][j] - so a SumAllSlots chain of 14
// rotate-and-add steps over 5488 ciphertexts is 14 launches + one permutation of c1 instead of 14 x 3 launches.
// Memory latency (one workgroup per CU - nobody else's arithmetic to hide behind; profiles/r05_ks14_pieces.txt: with every load switched
// off the kernel takes 23 ms per 5488-ciphertext link, with the plain loads 39.5): the source words of digit g + 1 are requested in four
// pieces while digit g is transformed - each piece at one pass boundary, taken out of the memory pipeline at the next, stage 0 applied and
// the result parked in the thread's own LDS slots (Ks14Next) - and the closing step issues every load of a component before its first store
// (out may alias the addends, so the compiler keeps loads behind earlier stores).
// Aliasing: a workgroup reads limb j of add0 / add1 / extra in full before (element-wise: at) the stores of limb j of out, and nothing
// else of those arrays - out may alias add0 / extra (in-place rotate-and-add); target and next_out must be separate arrays.
// LDS: the exchange image, then 64 KiB: the parked next digit during the digit loops; image + 64 KiB is the staging window afterwards.
#ifndef KS14_DBG
#define KS14_DBG 0          // timing experiments (tools/build_ks14_dbg.py; results are wrong with any bit set): 1 no c0 staging, 2 no accumulator, 4 no stash,
#endif                      // 8 no closing arithmetic, 16 synthetic digit sources (no loads), 32 synthetic keys (no loads)
#ifndef KS14_NT
#define KS14_NT 0           // non-temporal hints of the closing step (A/B, tools/build_ks14_nt.py): 1 result stores, 2 addend / accumulator loads, 4 the parked half (store + load), 8 the next link's c1
#endif
template <class P, class V> NTT_DEV void ks14_store(P *p, V v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }
template <class P> NTT_DEV auto ks14_load(const P *p, bool nt) -> typename std::remove_cv<P>::type { return nt ? __builtin_nontemporal_load(p) : *p; }
#ifndef KS14_PREFETCH
#define KS14_PREFETCH 1     // 0: every digit loads its source words at its start (A/B)
#endif
// stage 0 of the 2N'-point transform on one (x, y) = (c[e], c[e + N']) pair of source words, the output half `sg` keeps, recentred
// WHOLE: every source limb is ONE digit that covers it (2^dbc > q_l: the reference's N = 16384 networks, dbc 60) - the digit is the word itself, no 64-bit
// variable shift and mask per word (130 of ~2200 vector instructions per digit transform)
template <class AR, bool XI, bool WHOLE> struct Ks14Stage0 {
    typedef typename AR::T T;
    const DevConsts *C; typename AR::Mod m; double sg, w1; uint64_t mask;
    NTT_DEV T operator()(uint64_t sx, uint64_t sy, uint32_t l, int sh) const {
        if constexpr (XI) { const DMod ql = C->q[l]; const uint64_t xf = C->inv_qhat_q[l]; sx = mulmod(sx, xf, ql); sy = mulmod(sy, xf, ql); }   // digits of xi_l
        const T X = AR::from_u64(WHOLE ? sx : (sx >> sh) & mask), Y = AR::from_u64(WHOLE ? sy : (sy >> sh) & mask);
        return AR::center(__fma_rn(sg, AR::mulmod(Y, w1, m), X), m);         // x +- w y
    }
};
// the hook of ntt_forward_regs_hooked: piece c of the next digit's 16 pairs = registers 4c .. 4c+3, requested at boundary c, parked at boundary c + 1
// (Measured, not kept: the multiply-accumulate of a digit moved behind the NEXT digit's first pass - its 32 key words requested when that digit starts, the
// transformed values waiting in the thread's LDS slots meanwhile - to spend the key round trip (3.6 ms of 35 per link) under arithmetic: 37.8 vs 35.9 ms.)
template <class AR, bool XI, bool WHOLE> struct Ks14Next {
    typedef typename AR::T T;
    static constexpr uint32_t n2 = 8192, NT = 512;
    const Ks14Stage0<AR, XI, WHOLE> &st0;
    const uint64_t *src; uint32_t l; int sh;          // the next digit: limb, source limb index, shift (src == nullptr: there is none)
    T *park; uint32_t tid;
    uint64_t raw[8];
    template <int CH> NTT_DEV void issue() {
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t e = (uint32_t)(CH * 4 + q) * NT + tid; raw[2 * q] = src[e]; raw[2 * q + 1] = src[e + n2]; }
    }
    template <int CH> NTT_DEV void take() {
#pragma unroll
        for (int q = 0; q < 4; q++) park[(uint32_t)(CH * 4 + q) * NT + tid] = st0(raw[2 * q], raw[2 * q + 1], l, sh);
    }
    template <int PH> NTT_DEV void at() {
        if (!src) return;
        if constexpr (PH > 0) take<PH - 1>();
        if constexpr (PH < 4) issue<PH>();
    }
};
template <class AR, bool XI = false, bool WHOLE = false>
__global__ void __launch_bounds__(NttPlan<13>::NT) k_keyswitch_pair14(const uint64_t *__restrict__ target, size_t tgt_stride, const uint64_t *add0, const uint64_t *add1,
                                                                       size_t add_stride, const void *__restrict__ key_, uint64_t *out, double *__restrict__ stash,
                                                                       const DevConsts *__restrict__ C, int galois, uint32_t accmax, const uint64_t *extra, size_t ex_stride,
                                                                       uint64_t *const *__restrict__ out_tab, uint32_t perm_elt, uint32_t next_elt, uint64_t *__restrict__ next_out,
                                                                       uint32_t xcd_cts) {
    typedef typename AR::T T;
    static_assert(std::is_same<T, double>::value, "FP64 policies only");
    extern __shared__ __align__(16) unsigned char smem[];
    T *s = reinterpret_cast<T *>(smem);
    constexpr int L = 13;
    constexpr uint32_t n2 = 1u << L, n = 2 * n2, NT = NttPlan<L>::NT;
    constexpr int SA = NttPlan<L>::SA;
    static_assert(SA == 4 && NT == 512, "pass A pattern: register r of thread t holds coefficient r * 512 + t");
    constexpr uint32_t IMG = ntt_lds_words(n2) * 8;                   // bytes of the exchange image
    T *park = reinterpret_cast<T *>(smem + IMG);                       // [16][512]: the next digit, stage 0 applied, every thread its own slots
    // staging window of N words: the last 8 N bytes of (image + 64 KiB) - overlaps the image, used only between transforms
    uint64_t *win = reinterpret_cast<uint64_t *>(smem + (IMG + 65536 - (size_t)n * 8));
    const uint32_t k = C->k, tid = threadIdx.x;
    uint32_t ct = blockIdx.x / k, j = blockIdx.x % k;
    if (blockIdx.x < xcd_cts * k) { const uint32_t x = blockIdx.x & 7, sl = blockIdx.x >> 3; ct = 8 * (sl / k) + x; j = sl % k; }   // as in k_keyswitch_rr
    const DMod qm = C->q[j];
    const ArCtx<AR> A(C, j);
    typedef const NTT_GLOBAL double *GP;
    typedef typename KsPassA<AR, KS_SGPR_A != 0>::P FW;
    const int dbc = galois ? C->gdbc : C->dbc;
    const size_t kn = (size_t)k * n;
    const uint32_t tot = galois ? C->gk_tot : C->rl_tot;
    double *stp = stash + ((size_t)ct * k + j) * 2 * n2;
    const uint64_t *tgt = target + (size_t)ct * tgt_stride;
#pragma unroll 1
    for (uint32_t h = 0; h < 2; h++) {
        const GP fwg = (GP)(C->twdh + ((size_t)(j * 2 + 0) * 2 + h) * n2);
        typename FW::Tw fwh;
        fwh.w = fwg;
        if constexpr (HasPassA<FW>::value) ntt_load_pass_a<SA>(fwh, fwg);
        const typename AR::Tw ivh = {(GP)(C->twdh + ((size_t)(j * 2 + 1) * 2 + h) * n2)};
        const Ks14Stage0<AR, XI, WHOLE> st0{C, A.m, h ? -1.0 : 1.0, ntt_uniform(A.fw.w[1]), (1ull << dbc) - 1};
        T acc0[16], acc1[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { acc0[r] = 0; acc1[r] = 0; }
        const T *kp = reinterpret_cast<const T *>(key_) + (size_t)j * n + (size_t)h * n2;      // digit g: component 0 at kp + g 2kN, component 1 kN behind
        uint32_t terms = 0, l = 0, d = 0;
        for (uint32_t g = 0; g < tot; g++, kp += 2 * kn) {
            uint32_t tl = tid;
            asm volatile("" : "+v"(tl));               // as in k_keyswitch_rr: keep address math and twiddle loads inside the loop
            // the digit behind this one
            uint32_t ln = l, dn = d + 1;
            if (dn == (galois ? C->gk_dig[l] : C->rl_dig[l])) { ln = l + 1; dn = 0; }
            const bool pf = KS14_PREFETCH && !(KS14_DBG & 16);
            Ks14Next<AR, XI, WHOLE> nx{st0, (pf && g + 1 < tot) ? tgt + (size_t)ln * n : nullptr, ln, dbc * (int)dn, park, tl};
            T v[16];
            if (pf && g) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = park[(uint32_t)r * NT + tl];
            } else {
                const uint64_t *src = tgt + (size_t)l * n;
                const int sh = dbc * (int)d;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const uint32_t e = pass_index<L, SA, 0>(tl, r);
                    const uint64_t sx = (KS14_DBG & 16) ? (uint64_t)(e * 2654435761u + l) : src[e], sy = (KS14_DBG & 16) ? (uint64_t)(e * 40503u + l) : src[e + n2];
                    v[r] = st0(sx, sy, l, sh);
                }
            }
            ntt_forward_regs_hooked<FW, L, KS_PRE_SYNC != 0>(v, s, fwh, A.m, tl, nx);
            const T *k0 = kp, *k1 = kp + kn;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const uint32_t pos = tail_index<L>(tl, r);
                struct alignas(16) P2 { T a, b; };
                const P2 a = (KS14_DBG & 32) ? P2{v[r + 1], v[r]} : *reinterpret_cast<const P2 *>(k0 + pos), b = (KS14_DBG & 32) ? P2{v[r], v[r + 1]} : *reinterpret_cast<const P2 *>(k1 + pos);
                KsMac<AR>::mac(acc0[r], v[r], a.a, qm, A); KsMac<AR>::mac(acc0[r + 1], v[r + 1], a.b, qm, A);
                KsMac<AR>::mac(acc1[r], v[r], b.a, qm, A); KsMac<AR>::mac(acc1[r + 1], v[r + 1], b.b, qm, A);
            }
            nx.template at<4>();
            if (++terms == accmax) { terms = 0; KsMac<AR>::settle(acc0, A); KsMac<AR>::settle(acc1, A); }
            if (!KS_PRE_SYNC) __syncthreads();
            l = ln; d = dn;
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
            double *st = stp + (size_t)p * n2;
            if (h == 0) {                                              // parked: |u| <= q/2, this thread reads it back
#pragma unroll
                for (int r = 0; r < 16; r++) if (!(KS14_DBG & 4)) ks14_store(&st[pass_index<L, SA, 0>(tl, r)], AR::center(v[r], A.m), (KS14_NT & 4) != 0);
                __syncthreads();
                continue;
            }
            const uint64_t *ad = p ? add1 : add0;
            if (KS14_DBG & 1) ad = nullptr;
            if (ad) ad += (size_t)ct * add_stride + (size_t)j * n;
            const bool staged = ad && perm_elt && p == 0;
            const uint64_t *ex = extra && !(KS14_DBG & 2) ? extra + (size_t)ct * ex_stride + (size_t)p * kn + (size_t)j * n : nullptr;
            if (staged) {                                              // sigma(c0), limb j: N words through the window
                __syncthreads();                                       // everybody has taken its coefficients out of the image
#pragma unroll 8
                for (int r = 0; r < 32; r++) {
                    const uint32_t i = tl + NT * (uint32_t)r, pos = (i * perm_elt) & (2 * n - 1);
                    const uint64_t x = ad[i];
                    win[pos & (n - 1)] = (pos >> (L + 1)) ? negmod(x, qm.q) : x;
                }
                __syncthreads();
            }
            const double ni = C->ninvd[j], niw = AR::from_u64(C->ninv_w[j]);
            NTT_GLOBAL uint64_t *o = (NTT_GLOBAL uint64_t *)(out_tab ? out_tab[ct] : out + (size_t)ct * 2 * kn) + (size_t)p * kn + (size_t)j * n;
            const bool chain = next_elt && p == 1;
            if (chain) __syncthreads();                                // the window overlaps the image: everybody has taken its coefficients out
            // in two pieces of 8 registers; every load of a piece in front of its first store (out may alias the addends, so the compiler keeps a load behind
            // every earlier store - one exposed round trip per piece instead of one per register); addends summed as exact doubles (each below 2^50)
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                double u[8], alo[8], ahi[8];
#pragma unroll
                for (int r = 0; r < 8; r++) { u[r] = (KS14_DBG & 4) ? v[(r0 + r) ^ 1] : ks14_load(&st[pass_index<L, SA, 0>(tl, r0 + r)], (KS14_NT & 4) != 0); alo[r] = 0; ahi[r] = 0; }
                if (ex) {
#pragma unroll
                    for (int r = 0; r < 8; r++) { const uint32_t e = pass_index<L, SA, 0>(tl, r0 + r); alo[r] = AR::from_u64(ks14_load(&ex[e], (KS14_NT & 2) != 0)); ahi[r] = AR::from_u64(ks14_load(&ex[e + n2], (KS14_NT & 2) != 0)); }
                }
                if (staged) {
#pragma unroll
                    for (int r = 0; r < 8; r++) { const uint32_t e = pass_index<L, SA, 0>(tl, r0 + r); alo[r] = __dadd_rn(alo[r], AR::from_u64(win[e])); ahi[r] = __dadd_rn(ahi[r], AR::from_u64(win[e + n2])); }
                } else if (ad) {
#pragma unroll
                    for (int r = 0; r < 8; r++) { const uint32_t e = pass_index<L, SA, 0>(tl, r0 + r); alo[r] = __dadd_rn(alo[r], AR::from_u64(ks14_load(&ad[e], (KS14_NT & 2) != 0))); ahi[r] = __dadd_rn(ahi[r], AR::from_u64(ks14_load(&ad[e + n2], (KS14_NT & 2) != 0))); }
                }
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const uint32_t e = pass_index<L, SA, 0>(tl, r0 + r);
                    const double w = AR::center(v[r0 + r], A.m);
                    uint64_t lo = AR::to_u64(__dadd_rn(AR::mulmod(__dadd_rn(u[r], w), ni, A.m), alo[r]), A.m), hi = AR::to_u64(__dadd_rn(AR::mulmod(__dadd_rn(u[r], -w), niw, A.m), ahi[r]), A.m);
                    if (KS14_DBG & 8) { lo = (uint64_t)__double_as_longlong(u[r]); hi = (uint64_t)__double_as_longlong(v[r0 + r]); }
                    ks14_store(&o[e], lo, (KS14_NT & 1) != 0); ks14_store(&o[e + n2], hi, (KS14_NT & 1) != 0);
                    if (chain) {                                       // the new c1 limb once more, permuted for the next link of the chain
                        const uint32_t pl = (e * next_elt) & (2 * n - 1), ph = ((e + n2) * next_elt) & (2 * n - 1);
                        win[pl & (n - 1)] = (pl >> (L + 1)) ? negmod(lo, qm.q) : lo;
                        win[ph & (n - 1)] = (ph >> (L + 1)) ? negmod(hi, qm.q) : hi;
                    }
                }
            }
            __syncthreads();
            if (chain) {
                NTT_GLOBAL uint64_t *no = (NTT_GLOBAL uint64_t *)next_out + (size_t)ct * kn + (size_t)j * n;
#pragma unroll 8
                for (int r = 0; r < 32; r++) ks14_store(&no[tl + NT * (uint32_t)r], win[tl + NT * (uint32_t)r], (KS14_NT & 8) != 0);
            }
        }
    }
}
// 16 B of a key (two consecutive words) through a GLOBAL pointer (a struct cannot be copied out of address space 1; a vector type can)
template <class T> NTT_DEV void ks_load2(const NTT_GLOBAL T *p, T &a, T &b) {
    typedef T V2 __attribute__((ext_vector_type(2)));
    const V2 v = *reinterpret_cast<const NTT_GLOBAL V2 *>(p);
    a = v.x; b = v.y;
}
// A rotation's automorphism x -> x^elt applied while a two-launch key switch LOADS its operand, instead of a permutation kernel in front of
// it (k_galois_lds): one dispatch less per rotation - a single-image LoLa chain is ~230 dependent dispatches, 66 of them were this
// permutation, and the command processor retires ~5 us per dependent dispatch whatever queue it comes from.  The limb is
// loaded coalesced (coefficient tid + NT r, the first pass' own pattern), written to the exchange image at its destination
// (i elt mod 2N, negated when it wraps), and read back in the first pass' layout: raw[r] = sigma(a)[pass_index(tid, r)].
template <int L> NTT_DEV void ks_gather_automorphism(uint64_t (&raw)[16], const NTT_GLOBAL uint64_t *limb, uint32_t elt, uint64_t q, void *image, uint32_t tid) {
    constexpr uint32_t n = 1u << L, NT = NttPlan<L>::NT;
    uint64_t *s = reinterpret_cast<uint64_t *>(image);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t i = tid + NT * (uint32_t)r, pos = (i * elt) & (2 * n - 1);
        const uint64_t v = limb[i];
        s[lds_pos(pos & (n - 1))] = (pos >> L) ? (v ? q - v : 0) : v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) raw[r] = s[lds_pos(pass_index<L, NttPlan<L>::SA, 0>(tid, r))];
    __syncthreads();                                   // the image is the transform's again
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
                                                                  void *__restrict__ part_, const DevConsts *__restrict__ C, int galois, uint32_t tot, uint32_t perm_elt,
                                                                  const KsItem *__restrict__ items) {
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
    // (addresses that come out of a table are GLOBAL addresses: said so, the accesses stay global_load - a generic pointer would make them flat_load)
    const NTT_GLOBAL uint64_t *src = (const NTT_GLOBAL uint64_t *)target + (size_t)ct * tgt_stride + (size_t)l * n;
    const NTT_GLOBAL T *keyp = (const NTT_GLOBAL T *)key_;
    if (items) {                                           // rotations by per-ciphertext step counts in one launch: operand, key and Galois element from a table
        const KsItem it = items[ct];
        src = (const NTT_GLOBAL uint64_t *)it.in + kn + (size_t)l * n; keyp = (const NTT_GLOBAL T *)it.key; perm_elt = it.elt;
    }
    T v[16];
    uint64_t raw[16];
    if (perm_elt) ks_gather_automorphism<L>(raw, src, perm_elt, C->q[l].q, s, tid);
    else {
#pragma unroll
        for (int r = 0; r < 16; r++) raw[r] = src[pass_index<L, SA, 0>(tid, r)];
    }
    ks_premultiply(raw, C, l);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        uint64_t t = (raw[r] >> sh) & mask;
        if constexpr (std::is_same<T, uint64_t>::value) { if (mask >= q) t = t >= q ? bred128(t, 0, qm) : t; }
        v[r] = A.load(t);
    }
    if constexpr (std::is_same<T, double>::value) { if (mask >= q) AR::renorm(v, A.m); }
    ntt_forward_regs<AR, L>(v, s, A.fw, A.m, tid);
    const NTT_GLOBAL T *k0 = keyp + (size_t)g * 2 * kn + (size_t)j * n, *k1 = k0 + kn;
    T *o0 = reinterpret_cast<T *>(part_) + (((size_t)ct * tot + g) * 2) * kn + (size_t)j * n, *o1 = o0 + kn;
    struct alignas(16) P2 { T a, b; };
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const uint32_t pos = tail_index<L>(tid, r);
        P2 a, b;
        ks_load2<T>(k0 + pos, a.a, a.b); ks_load2<T>(k1 + pos, b.a, b.b);
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
                                                                 void *__restrict__ part_, const DevConsts *__restrict__ C, int galois, uint32_t accmax, uint32_t perm_elt,
                                                                 const KsItem *__restrict__ items) {
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
    const NTT_GLOBAL uint64_t *src = (const NTT_GLOBAL uint64_t *)target + (size_t)ct * tgt_stride + (size_t)l * n;
    const NTT_GLOBAL T *keyp = (const NTT_GLOBAL T *)key_;
    if (items) {
        const KsItem it = items[ct];
        src = (const NTT_GLOBAL uint64_t *)it.in + kn + (size_t)l * n; keyp = (const NTT_GLOBAL T *)it.key; perm_elt = it.elt;
    }
    uint64_t raw[16];
    if (perm_elt) ks_gather_automorphism<L>(raw, src, perm_elt, C->q[l].q, s, tid);
    else {
        uint32_t t0 = tid;
        asm volatile("" : "+v"(t0));
#pragma unroll
        for (int r = 0; r < 16; r++) raw[r] = src[pass_index<L, SA, 0>(t0, r)];
    }
    ks_premultiply(raw, C, l);
    T acc0[16], acc1[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0; acc1[r] = 0; }
    const NTT_GLOBAL T *kp = keyp + (size_t)g0 * 2 * kn;
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
        const NTT_GLOBAL T *k0 = kp + (size_t)j * n, *k1 = kp + kn + (size_t)j * n;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const uint32_t pos = tail_index<L>(tl, r);
            struct alignas(16) P2 { T a, b; };
            P2 a, b;
            ks_load2<T>(k0 + pos, a.a, a.b); ks_load2<T>(k1 + pos, b.a, b.b);
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
                                                                 uint32_t accmax, const uint64_t *extra, size_t ex_stride, uint64_t *const *__restrict__ out_tab, uint32_t perm_elt,
                                                                 const KsItem *__restrict__ items) {
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
    const NTT_GLOBAL uint64_t *ad = (const NTT_GLOBAL uint64_t *)(p ? add1 : add0);
    if (ad) ad += (size_t)ct * add_stride + (size_t)j * n;
    if (items) { const KsItem it = items[ct]; ad = p ? nullptr : (const NTT_GLOBAL uint64_t *)it.in + (size_t)j * n; perm_elt = it.elt; }
    uint64_t addv[16];
    if (ad && perm_elt) {                                  // sigma(c0): the whole limb is read (through the image) before a word of the result is written
        __syncthreads();                                   // everybody has taken its coefficients out of the image
        ks_gather_automorphism<L>(addv, ad, perm_elt, qm.q, s, tid);
    } else if (ad) {
#pragma unroll
        for (int r = 0; r < 16; r++) addv[r] = ad[pass_index<L, SA, 0>(tid, r)];
    }
    NTT_GLOBAL uint64_t *o = (NTT_GLOBAL uint64_t *)(out_tab ? out_tab[ct] : out + (size_t)ct * 2 * kn) + (size_t)p * kn + (size_t)j * n;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t e = pass_index<L, SA, 0>(tid, r);
        uint64_t val = A.scaled(v[r]);
        if (ad) val = addmod(val, addv[r], qm.q);
        if (extra) val = addmod(val, extra[(size_t)ct * ex_stride + (size_t)p * kn + (size_t)j * n + e], qm.q);
        o[e] = val;
    }
}

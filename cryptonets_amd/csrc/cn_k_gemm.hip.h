// HOT LOOP A: scalar GEMM kernels (AtomicSealBfvVector.cs:434-521).  Included by cn_l_gemm.hip only.
#pragma once
#include "cn_dev_common.hip.h"

// ------------------------------------------------------------------ HOT LOOP A: scalar GEMM
// Workgroup -> (coefficient chunk, limb, output tile mt, group g).  The mtiles output tiles of one (chunk, limb, g) read the SAME
// input elements; workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so the tiles take ids b, b+8, b+16, ...
// - same XCD, adjacent in time: the K input elements are fetched from HBM once and re-read from that XCD's L2 (dense 845->100:
// 5 tiles; without this every tile streamed the 528 MB of input ciphertexts again).
// order 1 (slice-major): a SLICE = (limb, chunk of 256 coefficients) of every ciphertext is 2 KiB per ciphertext - all 784 inputs of the CryptoNets
// convolution are 1.5 MiB, which an XCD's 4 MiB L2 holds.  XCD x (blocks b = x mod 8) walks through ALL groups of slice x, then of slice x + 8, ...:
// every input slice is fetched once per launch, whatever the overlap of the gather lists (5x5 windows at stride 2 share each pixel 6.25 times; in
// group-major order only the horizontal overlap survived in L2: FETCH_SIZE 2.3 x the input bytes, profiles/r02_pmc_square_gemm.txt).
// order 1 is launched as a 2-D grid (8 * mtiles * G, slices / 8) - x fastest, so workgroup (x, y) still lands on XCD x mod 8: the coordinates of the
// convolution's workgroups (mtiles = 1; chunks is a power of two) come out of shifts and masks.  (As one linear id they took three divisions by run-time
// values - ~100 vector instructions of the ~860 a convolution workgroup issues, on a kernel that is bound by its issue slots.)
DEV void gemm_block_coords(uint32_t bx, uint32_t by, uint32_t chunks, uint32_t limbs, uint32_t mtiles, uint32_t G, uint32_t &chunk, uint32_t &limb, uint32_t &mt, uint32_t &g,
                           uint32_t order = 0) {
    if (order == 1) {
        const uint32_t x = bx & 7, r = bx >> 3;
        if (mtiles == 1) { mt = 0; g = r; } else { mt = r % mtiles; g = r / mtiles; }
        const uint32_t sl = by * 8 + x;
        chunk = sl & (chunks - 1); limb = sl >> (__ffs((int)chunks) - 1);
        return;
    }
    const uint32_t b = bx, D = chunks * limbs * G;
    uint32_t d;
    if ((D & 7) == 0) { const uint32_t r = b >> 3; mt = r % mtiles; d = (r / mtiles) * 8 + (b & 7); }
    else { d = b % D; mt = b / D; }
    chunk = d % chunks; limb = (d / chunks) % limbs; g = d / (chunks * limbs);
}
// Group g gathers K input ciphertexts idx[g][:] once and produces M outputs (register tile MT):
//   out[out_idx[g][m]] = sum_k Wl[j][g][m][k] * in[idx[g][k]]  (+ scaled bias)   per limb j, coefficient i.
// Products accumulate lazily in 128 bits; one Barrett reduction per `lazy` terms.
// ABS: the gather / output / bias tables hold DEVICE ADDRESSES (u64, 0 = padded tap / no output / no bias) instead of indices relative
// to `in` / `out` / `bias` - the form the deferred per-ciphertext calls arrive in, where every ciphertext is its own array.
// A device address read from a table is a GLOBAL address: said so, the access is global_load / global_store.  As a generic pointer it is
// flat_load - counted on lgkmcnt as well, so every wait for a scalar weight load also drained the input loads in flight (the address-table
// kernels of the deferred calls ran 30-60 % behind their index-table twins).
DEV const NTT_GLOBAL uint64_t *gmem(uint64_t a) { return (const NTT_GLOBAL uint64_t *)a; }
DEV NTT_GLOBAL uint64_t *gmem_w(uint64_t a) { return (NTT_GLOBAL uint64_t *)a; }
// Index units of the table-driven (non-ABS) kernels, in words: input / output index x unit = word offset from `in` / `out`, bias index x unit from `bias`.  0 = the defaults (one
// ciphertext / one plaintext per index: the arrays of a plan).  The deferred per-ciphertext calls - every ciphertext its own array - hand in 32-bit offsets in units of 32 words
// (256 B: the allocation granule) from the lowest address of their flush group (round 6): the same index-table kernels as a plan instead of 64-bit address tables.
struct GemmUnits { uint32_t in, out, bias; };
template <bool ABS> struct GemmTab { typedef int32_t T; };
template <> struct GemmTab<true> { typedef uint64_t T; };
template <int MT, bool ABS = false>
__global__ void __launch_bounds__(256) k_scalar_gemm(const uint64_t *__restrict__ in, const void *__restrict__ idx_, const uint64_t *__restrict__ Wl,
                                                     const void *__restrict__ out_idx_, const uint64_t *__restrict__ bias, const void *__restrict__ bias_idx_,
                                                     uint64_t *__restrict__ out, const DevConsts *__restrict__ C, uint32_t chunks, uint32_t G, uint32_t M,
                                                     uint32_t K, uint32_t mtiles, uint32_t lazy, uint32_t Kp, uint32_t obase, uint32_t polys, uint32_t order, GemmUnits U) {
    typedef typename GemmTab<ABS>::T TT;
    const TT *idx = (const TT *)idx_, *out_idx = (const TT *)out_idx_, *bias_idx = (const TT *)bias_idx_;
    const uint32_t n = C->n, k = C->k, limbs = polys * k;       // polys = ciphertext size: 2, or 3 for unrelinearized products (Evaluator::multiply_plain / add accept both)
    uint32_t chunk, limb, mt, g;
    gemm_block_coords(blockIdx.x, blockIdx.y, chunks, limbs, mtiles, G, chunk, limb, mt, g, order);
    const uint32_t j = limb % k, i = chunk * blockDim.x + threadIdx.x;
    const size_t ctw = (size_t)limbs * n, e = (size_t)limb * n + i;
    const size_t iu = U.in ? U.in : ctw, ou = U.out ? U.out : ctw, bu = U.bias ? U.bias : n;
    const DMod qm = C->q[j];
    u128 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = 0;
    const TT *gi = idx + (size_t)g * Kp;                                                 // row pitch Kp: 16 B aligned, -1 (ABS: 0) beyond K
    const uint64_t *gw = Wl + (((size_t)j * G + g) * mtiles + mt) * (size_t)K * MT;      // [kk][m]: the MT weights of a term are contiguous
    const uint32_t mcnt = min((uint32_t)MT, M - mt * MT);
    for (uint32_t k0 = 0; k0 < K; k0 += lazy) {            // reduction between blocks of `lazy` terms (see k_scalar_gemm_f64)
        const uint32_t k1 = min(K, k0 + lazy);
        for (uint32_t kk = k0; kk < k1; kk++) {
            uint64_t x;
            if constexpr (ABS) { const uint64_t a = gi[kk]; if (!a) continue; x = gmem(a)[e]; }
            else { const int32_t id = gi[kk]; if (id < 0) continue; x = in[(size_t)id * iu + e]; }
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
        const uint32_t o = g * M + mt * MT + m;
        // bias: a PoolLayer bias is the constant polynomial (BatchEncoder of a constant vector) - every coefficient but one is zero and
        // scales to zero: the 128-bit multiply + Barrett of scale_plain is skipped there (wave-uniformly almost everywhere)
        if constexpr (ABS) {
            if ((uint32_t)m < mcnt && out_idx[o]) {
                uint64_t r = bred128(acc[m], qm);
                if (bias_idx && bias_idx[o] && limb < k) { const uint64_t bv = gmem(bias_idx[o])[i]; if (bv) r = addmod(r, scale_plain(C, bv, j), qm.q); }
                gmem_w(out_idx[o])[e] = r;
            }
        } else if ((uint32_t)m < mcnt && out_idx[o] >= 0) {                     // -1: padding member of a smaller group
            uint64_t r = bred128(acc[m], qm);
            if (bias && limb < k) { const uint64_t bv = bias[(size_t)bias_idx[o] * bu + i]; if (bv) r = addmod(r, scale_plain(C, bv, j), qm.q); }
            out[(size_t)(obase + (uint32_t)out_idx[o]) * ou + e] = r;
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
// limb l of a canonical residue (below 2^(NL LW)) as an exact double; the top limb needs no mask.  NL = 1: the WHOLE word (v | 2^52's exponent, minus 2^52: two
// instructions) - the launcher picks that form when every row's sum of |weights| times q_max stays below 2^53 (gemm_one_limb): one FMA per MAC and nothing to fold
template <int NL, int LW> DEV double gemm_limb(uint64_t x, int l) {
    if constexpr (NL == 1) return BzF::from_u64(x & 0x000FFFFFFFFFFFFFull);   // (masked: a padded tap multiplies whatever word it reads by the weight 0 - an unwritten array
                                                                               // may hold bits 52..62, which would land in the exponent: 0 x NaN poisons the sum; one v_and_b32 per term)
    else return (double)(uint32_t)(l == NL - 1 ? x >> (l * LW) : (x >> (l * LW)) & ((1ull << LW) - 1));
}
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
template <int MT, int NL, int LW, bool ABS = false>
__global__ void __launch_bounds__(256) k_scalar_gemm_f64(const uint64_t *__restrict__ in, const void *__restrict__ idx_, const double *__restrict__ Wd,
                                                         const void *__restrict__ out_idx_, const uint64_t *__restrict__ bias, const void *__restrict__ bias_idx_,
                                                         uint64_t *__restrict__ out, const DevConsts *__restrict__ C, uint32_t chunks, uint32_t G, uint32_t M,
                                                         uint32_t K, uint32_t mtiles, uint32_t lazy, uint32_t Kp, uint32_t obase, uint32_t polys, uint32_t order, uint32_t Kw, GemmUnits U) {
    typedef typename GemmTab<ABS>::T TT;
    const TT *idx = (const TT *)idx_, *out_idx = (const TT *)out_idx_, *bias_idx = (const TT *)bias_idx_;
    const uint32_t n = C->n, k = C->k, limbs = polys * k;       // polys = ciphertext size: 2, or 3 for unrelinearized products (Evaluator::multiply_plain / add accept both)
    uint32_t chunk, limb, mt, g;
    gemm_block_coords(blockIdx.x, blockIdx.y, chunks, limbs, mtiles, G, chunk, limb, mt, g, order);
    const uint32_t j = limb % k, i = chunk * blockDim.x + threadIdx.x;
    const size_t ctw = (size_t)limbs * n, e = (size_t)limb * n + i;
    const size_t iu = U.in ? U.in : ctw, ou = U.out ? U.out : ctw, bu = U.bias ? U.bias : n;
    const DMod qm = C->q[j];
    double acc[NL][MT], res[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        res[m] = 0.0;
#pragma unroll
        for (int l = 0; l < NL; l++) acc[l][m] = 0.0;
    }
    const TT *gi = idx + (size_t)g * Kp;                                                 // row pitch Kp: 16 B aligned, -1 (ABS: 0) beyond K
    const double *gw = Wd + ((size_t)g * mtiles + mt) * (size_t)Kw * MT;                 // [kk < Kw][m]: zero rows behind the K terms (gemm_f64_rows)
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
    // current PF terms are issued.  The requests are branch-free (padded taps and terms past K multiply a valid word by the weight 0: every
    // (group, tile) of the table ends in zero rows, gemm_f64_rows), so the waits stay exact.
#ifndef GEMM_PF_SMALL
#define GEMM_PF_SMALL 4       // terms requested ahead per register set for MT <= 5 (convolution windows: 25 taps, 5 maps).  8 - two gather lists of four in flight per
                              // thread, VERDICT r03 next #7 - was measured in round 4 and is NOT the default: 91 instead of 66 VGPRs (5 instead of 7 waves per SIMD),
                              // 476 against 462 us for the CryptoNets convolution (profiles/HISTORY.md, round 4); again on the round-5 kernel (80 VGPRs, 26 spilled
                              // SGPRs for the weights of a set): 451-455 against 382-386 us
#endif
    constexpr int PF = (MT <= 5) ? GEMM_PF_SMALL : 4;
    static_assert(PF % 4 == 0, "gather indices travel in 16 B scalar loads of four");
    // A padded tap (-1 / address 0 in the gather row) and a term past K carry the WEIGHT 0 in the table (pack_gemm_weights) and read a valid word (element
    // 0 / the fallback ciphertext); a block of `lazy` terms ends on a multiple of 2 PF unless it ends at K: no per-lane select on the 64-bit words.
    auto fetch = [&](uint64_t (&x)[PF], uint32_t kk) {
        // ONE 16 B scalar load per four gather indices (four dependent s_load_dword + wait chains cost more than the FMAs)
        const uint32_t kc = min(kk, Kp - PF);
        if constexpr (ABS) {               // addresses: two 16 B scalar loads per four; a padded tap reads (and multiplies by 0) the fallback ciphertext `in`
#pragma unroll
            for (int c = 0; c < PF; c += 4) {
                const ulonglong2 a01 = *reinterpret_cast<const ulonglong2 *>(__builtin_assume_aligned(gi + kc + c, 16));
                const ulonglong2 a23 = *reinterpret_cast<const ulonglong2 *>(__builtin_assume_aligned(gi + kc + c + 2, 16));
                const uint64_t ad[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
                for (int p = 0; p < 4; p++) x[c + p] = gmem(ad[p] ? ad[p] : (uint64_t)in)[e];
            }
        } else {
#pragma unroll
            for (int c = 0; c < PF; c += 4) {
                const int4 ids = *reinterpret_cast<const int4 *>(__builtin_assume_aligned(gi + kc + c, 16));
                const int32_t id[4] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
                for (int p = 0; p < 4; p++) x[c + p] = in[(size_t)max(id[p], 0) * iu + e];
            }
        }
    };
    // the weights of WB terms (up to 20 doubles = 40 SGPRs) are contiguous: requested together (wide scalar loads, ONE wait) before the first product.
    // WSET (the whole set fits: MT <= 5): they are requested BEFORE the gather entries of the set behind it (wload in front of fetch), so that the one
    // lgkmcnt(0) in front of that set's input requests covers them too - one scalar round trip per set on a wave's chain of dependent requests instead of two
    constexpr int WB = MT >= 20 ? 1 : (MT * PF <= 20 ? PF : 20 / MT);
    constexpr bool WSET = WB == PF;
    auto wload = [&](double (&w)[WB * MT], uint32_t kk) {
        const double *wp = gw + (size_t)kk * MT;
#pragma unroll
        for (int q = 0; q < WB * MT; q++) w[q] = wp[q];
    };
    auto terms = [&](const uint64_t (&x)[PF], uint32_t kk, const double (&w0)[WB * MT]) {
#pragma unroll
        for (int p0 = 0; p0 < PF; p0 += WB) {
            double w[WB * MT];
            if constexpr (WSET) {
#pragma unroll
                for (int q = 0; q < WB * MT; q++) w[q] = w0[q];
            } else wload(w, kk + p0);
#pragma unroll
            for (int p = p0; p < p0 + WB; p++) {
                double xl[NL];
#pragma unroll
                for (int l = 0; l < NL; l++) xl[l] = gemm_limb<NL, LW>(x[p], l);
#pragma unroll
                for (int m = 0; m < MT; m++) {
#pragma unroll
                    for (int l = 0; l < NL; l++) acc[l][m] = __fma_rn(xl[l], w[(p - p0) * MT + m], acc[l][m]);
                }
            }
        }
    };
    constexpr bool DPPW = GEMM_DPP_W && MT == 20 && PF == 4;
    constexpr int WV = DPPW ? PF * MT / 16 : 1;                    // 4 terms x 20 weights = 5 registers of 16 lanes
    auto wfetch = [&](double (&wv)[WV], uint32_t kk) {             // rows min(kk, K + 4) .. + 3: inside the zero rows behind the K terms
        const double *wp = gw + (size_t)min(kk, K + 4) * MT + (threadIdx.x & 15);
#pragma unroll
        for (int v = 0; v < WV; v++) wv[v] = wp[16 * v];
    };
    auto terms_dpp = [&](const uint64_t (&x)[PF], double (&wv)[WV]) {
#pragma unroll
        for (int p = 0; p < PF; p++) {
            double xl[NL];
#pragma unroll
            for (int l = 0; l < NL; l++) xl[l] = gemm_limb<NL, LW>(x[p], l);
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
            fetch(xa, k0); wfetch(wa, k0);
            for (uint32_t kk = k0; kk < k1; kk += 2 * PF) {
                fetch(xb, kk + PF); wfetch(wb, kk + PF);
                terms_dpp(xa, wa);
                fetch(xa, kk + 2 * PF); wfetch(wa, kk + 2 * PF);
                terms_dpp(xb, wb);
            }
        } else {
            // sets of PF terms in pairs, then the odd set: a window of 25 taps is 7 sets of four, not 8.  ONE exit at the end of the pair loop and no branch
            // inside it - at a join the compiler's wait counts assume the path with the fewest requests in flight and drain the set requested ahead.
            const uint32_t sets = (k1 - k0 + PF - 1) / PF;
            uint32_t kk = k0;
            double wa[WB * MT], wb[WB * MT];
            fetch(xa, k0);
            for (uint32_t it = sets >> 1; it; it--, kk += 2 * PF) {
                if constexpr (WSET) { wload(wa, kk); __builtin_amdgcn_sched_barrier(0); }      // (the scheduler sinks the request behind the input requests otherwise)
                fetch(xb, kk + PF);
                terms(xa, kk, wa);
                if constexpr (WSET) { wload(wb, kk + PF); __builtin_amdgcn_sched_barrier(0); }
                fetch(xa, kk + 2 * PF);
                terms(xb, kk + PF, wb);
            }
            if (sets & 1) { if constexpr (WSET) wload(wa, kk); terms(xa, kk, wa); }
        }
        fold();
    }
    // Outputs in sets of OB: the table entries of a set (scalar loads), then its bias words (all requested before the first is used), then the arithmetic and
    // the stores - one exposed round trip per kind and set.  (One output at a time it was three dependent round trips per output: table entry -> bias entry ->
    // bias word, ~8 us of a convolution workgroup's ~19 us.)
    constexpr int OB = MT < 5 ? MT : 5;
    const uint32_t olast = G * M - 1;
    const bool hb = (ABS ? bias_idx != nullptr : bias != nullptr) && limb < k;
#pragma unroll
    for (int m0 = 0; m0 < MT; m0 += OB) {
        uint32_t oo[OB];
        TT oi[OB], bi[OB];
        bool st[OB];
        uint64_t bv[OB];
#pragma unroll
        for (int u = 0; u < OB; u++) oo[u] = min(g * M + mt * MT + m0 + u, olast);
#pragma unroll
        for (int u = 0; u < OB; u++) { oi[u] = out_idx[oo[u]]; bi[u] = 0; bv[u] = 0; }
        if (hb) {
#pragma unroll
            for (int u = 0; u < OB; u++) bi[u] = bias_idx[oo[u]];
        }
#pragma unroll
        for (int u = 0; u < OB; u++) st[u] = (uint32_t)(m0 + u) < mcnt && (ABS ? oi[u] != 0 : (int64_t)oi[u] >= 0);
        if (hb) {
#pragma unroll
            for (int u = 0; u < OB; u++) {
                if constexpr (ABS) bv[u] = gmem(st[u] && bi[u] ? bi[u] : (uint64_t)in)[i];          // no bias: any readable word, discarded below
                else bv[u] = bias[(size_t)(st[u] ? bi[u] : (TT)0) * bu + i];
            }
#pragma unroll
            for (int u = 0; u < OB; u++) if (!(st[u] && (ABS ? bi[u] != 0 : true))) bv[u] = 0;
        }
#pragma unroll
        for (int u = 0; u < OB; u++) {
            if (!st[u]) continue;
            // res left the last fold recentred (|res| <= q/2 + 1 ulp of the quotient): canonical residue without a second quotient
            const double rc = res[m0 + u] < 0.0 ? __dadd_rn(res[m0 + u], mq.q) : res[m0 + u];
            uint64_t r = (uint64_t)__double_as_longlong(__dadd_rn(rc, 4503599627370496.0)) & 0x000FFFFFFFFFFFFFull;
            // bias: a PoolLayer bias is the constant polynomial - every coefficient but one is zero and scales to zero (skipped wave-uniformly almost everywhere)
            if (bv[u]) r = addmod(r, scale_plain(C, bv[u], j), qm.q);
            if constexpr (ABS) gmem_w(oi[u])[e] = r;
            else out[(size_t)(obase + (uint32_t)oi[u]) * ou + e] = r;
        }
    }
}

// ------------------------------------------------------------------ scalar GEMM on the int8 matrix cores (exact)
// out[o][e] = sum_k W[o][k] * x[k][e] mod q_j is an integer contraction over K input ciphertexts: M = outputs, N = coefficients, K = inputs.
// With SMALL signed weights it maps onto v_mfma_i32_32x32x32_i8 EXACTLY: the residue x < 2^46 is recoded into 6 signed base-256 digits
// (x + 0x80..80 has bytes b_i; b_i - 128 = b_i ^ 0x80 as int8), the weight into P <= 3 signed digits the same way; digit x digit products
// (<= 2^14) accumulate in i32 over K P < 2^17 terms without overflow, products of equal weight 256^(i+p) share one accumulator
// ("diagonal"), and the 6 + P - 1 diagonals are folded mod q_j in exact FP64 at the end.
// Workgroup = 4 waves = 32 coefficient columns x up to four 32-row output tiles (wave w: m-tile 4 mg + w).  The B operand (digits of 32
// inputs x 32 columns per K step) is the same for the four waves, so they build it TOGETHER: wave w loads the input words of K slots
// 4w .. 4w+3 of its lanes' half (4 loads of 256 contiguous bytes per half-wave instead of 16), recodes them, transposes their bytes with
// v_perm_b32 into one dword per digit plane, and publishes the six dwords in LDS; after one barrier every wave reads its complete
// fragments back as 16 B per lane and plane.  Three LDS buffers and a ring of three register sets keep one barrier per K step with the input words and
// the A fragments (weight digits, laid out in fragment order by the plan: one 16 B load per lane and plane) of the next TWO steps in flight; the gather-table
// entries of a wave's slots are wave-uniform and arrive through the scalar cache, a step ahead of the loads that use them (round 5).
// K-slot convention: lane l, byte t of an operand <-> k = 32 ks + 16 (l >> 5) + t for BOTH operands (the instruction pairs equal slots,
// so any convention shared by A and B is correct).  C/D: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
DEV uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }   // byte n of the result = byte sel[n] of {hi: 4..7, lo: 0..3}
#ifndef GEMM_MFMA_DEPTH
#define GEMM_MFMA_DEPTH 2          // K steps of input words in flight beside the one being multiplied (1 or 2)
#endif
template <int P, bool ABS>
__global__ void __launch_bounds__(256, 2) k_scalar_gemm_mfma(const uint64_t *__restrict__ in, const void *__restrict__ idx_, const int8_t *__restrict__ Wf,
                                                             const void *__restrict__ out_idx_, const uint64_t *__restrict__ bias, const void *__restrict__ bias_idx_,
                                                             uint64_t *__restrict__ out, const DevConsts *__restrict__ C, uint32_t G, uint32_t M, uint32_t mtiles,
                                                             uint32_t ksteps, uint32_t obase, uint32_t polys, GemmUnits U) {
    typedef typename GemmTab<ABS>::T TT;
    const TT *idx = (const TT *)idx_, *out_idx = (const TT *)out_idx_, *bias_idx = (const TT *)bias_idx_;
    constexpr int ND = 6, D = ND + P - 1, NB = GEMM_MFMA_DEPTH + 1;
    __shared__ __align__(16) uint32_t frag[NB][ND][64][4];         // [buffer][digit plane][lane][slot group q]: 6 KiB each
    const uint32_t n = C->n, k = C->k, limbs = polys * k, ctiles = n >> 5;
    // the wave number as a SCALAR: the gather-table entries of a wave's K slots are then wave-uniform and travel through the scalar cache (lgkmcnt).  As vector
    // loads they sat on vmcnt IN FRONT of the input loads that need them: every step waited for the table (vmcnt(0) - which also drained whatever input words
    // were still in flight), then for the inputs: two exposed round trips per K step, 1.9 us of them against 0.2 us of matrix instructions.
    const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), half = lane >> 5, col = lane & 31;
    uint32_t b = blockIdx.x;
    const uint32_t ctile = b % ctiles; b /= ctiles;
    const uint32_t limb = b % limbs; b /= limbs;
    const uint32_t mgroups = (mtiles + 3) >> 2, mg = b % mgroups, g = b / mgroups;
    const uint32_t mt = mg * 4 + wave;
    const bool active = mt < mtiles;                           // a wave without an output tile still loads its share of the B operand
    const uint32_t j = limb % k;
    const size_t ctw = (size_t)limbs * n, e = (size_t)limb * n + (size_t)ctile * 32 + col;
    const size_t iu = U.in ? U.in : ctw, ou = U.out ? U.out : ctw, bu = U.bias ? U.bias : n;
    const uint32_t Kp = ksteps * 32;
    const TT *gi = idx + (size_t)g * Kp + 4 * wave;            // + 32 ks + 16 half + u
    const int8_t *wf = Wf + ((((size_t)g * P) * mtiles + (active ? mt : 0)) * ksteps) * 1024 + (size_t)lane * 16;       // + (p * mtiles * ksteps + ks) * 1024
    v16i_t acc[D];
#pragma unroll
    for (int d = 0; d < D; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[d][r] = 0;
    const uint64_t BIAS = 0x0000808080808080ull;
    // Requests behind the last step are NOT skipped but repeat the last step (a few words re-read from L2): with a branch around the loads the compiler's wait
    // counts at the join assume the path WITHOUT them - "at most 5 younger loads" where 11 are in flight - and every step drained the sets requested ahead.
    const uint32_t klast = ksteps - 1;
    auto load_idx = [&](TT (&s)[8], uint32_t ks) {              // table entries of slots 4 wave .. + 3 (lanes 0..31) and 16 + 4 wave .. + 3 (lanes 32..63): scalar loads
        const TT *t = gi + (size_t)min(ks, klast) * 32;
#pragma unroll
        for (int u = 0; u < 4; u++) { s[u] = t[u]; s[4 + u] = t[16 + u]; }
    };
    auto load_x = [&](uint64_t (&x)[4], const TT (&s)[8]) {     // the 4 input words of this lane's slots
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if constexpr (ABS) {
                const uint64_t a0 = s[u] ? s[u] : (uint64_t)in, a1 = s[4 + u] ? s[4 + u] : (uint64_t)in;      // padded tap: any readable word (its weight digits are 0)
                x[u] = gmem(half ? a1 : a0)[e];
            } else {
                // both products on the scalar unit, THEN the per-lane choice (left alone the compiler chooses first and multiplies per lane: two quarter-rate
                // v_mad_u64_u32 per word)
                size_t o0 = (size_t)max(s[u], 0) * iu, o1 = (size_t)max(s[4 + u], 0) * iu;
                asm volatile("" : "+s"(o0), "+s"(o1));
                x[u] = in[(half ? o1 : o0) + e];
            }
        }
    };
    auto load_a = [&](v4i_t (&af)[P], uint32_t ks) {
#pragma unroll
        for (int p = 0; p < P; p++) af[p] = *reinterpret_cast<const v4i_t *>(wf + ((size_t)p * mtiles * ksteps + min(ks, klast)) * 1024);
    };
    // x / af: NB register sets in a ring - set (ks mod NB) holds step ks, requested GEMM_MFMA_DEPTH steps ahead; `nidx` = the table entries of the step whose
    // inputs are requested next (asked for one step before they are needed; refilled right after use)
    uint64_t xs[NB][4];
    v4i_t afs[NB][P];
    TT nidx[8];
    load_idx(nidx, 0);
#pragma unroll
    for (int s = 0; s < GEMM_MFMA_DEPTH; s++) { load_x(xs[s], nidx); load_a(afs[s], s); load_idx(nidx, s + 1); }
    auto step = [&](uint64_t (&x)[4], v4i_t (&af)[P], uint64_t (&xn)[4], v4i_t (&afn)[P], uint32_t ks, uint32_t (*fb)[64][4]) {
        // recode: byte i of y = signed digit i; transpose the 4 words into one dword per digit plane; publish
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint64_t y = (x[u] + BIAS) ^ BIAS; lo[u] = (uint32_t)y; hi[u] = (uint32_t)(y >> 32); }
        const uint32_t ab02 = byte_perm(lo[1], lo[0], 0x06020400u), cd02 = byte_perm(lo[3], lo[2], 0x06020400u);     // (a0 b0 a2 b2), (c0 d0 c2 d2)
        const uint32_t ab13 = byte_perm(lo[1], lo[0], 0x07030501u), cd13 = byte_perm(lo[3], lo[2], 0x07030501u);     // (a1 b1 a3 b3), (c1 d1 c3 d3)
        const uint32_t hab = byte_perm(hi[1], hi[0], 0x05010400u), hcd = byte_perm(hi[3], hi[2], 0x05010400u);       // (a4 b4 a5 b5), (c4 d4 c5 d5)
        fb[0][lane][wave] = byte_perm(cd02, ab02, 0x05040100u);          // a0 b0 c0 d0
        fb[1][lane][wave] = byte_perm(cd13, ab13, 0x05040100u);
        fb[2][lane][wave] = byte_perm(cd02, ab02, 0x07060302u);          // a2 b2 c2 d2
        fb[3][lane][wave] = byte_perm(cd13, ab13, 0x07060302u);
        fb[4][lane][wave] = byte_perm(hcd, hab, 0x05040100u);
        fb[5][lane][wave] = byte_perm(hcd, hab, 0x07060302u);
        load_x(xn, nidx); load_a(afn, ks + GEMM_MFMA_DEPTH);             // in flight over the barriers and matrix instructions of the next GEMM_MFMA_DEPTH steps
        load_idx(nidx, ks + GEMM_MFMA_DEPTH + 1);
        __syncthreads();
        if (active) {
            v4i_t bf[ND];
#pragma unroll
            for (int i = 0; i < ND; i++) bf[i] = *reinterpret_cast<const v4i_t *>(&fb[i][lane][0]);
            // P x 6 matrix instructions; equal digit weights share an accumulator
#pragma unroll
            for (int p = 0; p < P; p++)
#pragma unroll
                for (int i = 0; i < ND; i++) acc[i + p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[p], bf[i], acc[i + p], 0, 0, 0);
        }
    };
    // the set that step ks refills is the one step ks - 1 consumed
    // (for the same reason the loop has ONE exit, at its end: whole turns of the ring, then the ksteps mod NB steps left over)
    uint32_t ks = 0;
    for (; ks + NB <= ksteps; ks += NB) {
#pragma unroll
        for (int s = 0; s < NB; s++) step(xs[s], afs[s], xs[(s + GEMM_MFMA_DEPTH) % NB], afs[(s + GEMM_MFMA_DEPTH) % NB], ks + s, frag[s]);
    }
#pragma unroll
    for (int s = 0; s < NB - 1; s++)
        if (ks + s < ksteps) step(xs[s], afs[s], xs[(s + GEMM_MFMA_DEPTH) % NB], afs[(s + GEMM_MFMA_DEPTH) % NB], ks + s, frag[s]);
    if (!active) return;
    // ---- fold the diagonals: value = sum_d acc_d 256^d mod q_j (exact FP64), bias, store
    const BzF::Mod mq = {C->qd[j], C->qinvd[j]};
    const DMod qm = C->q[j];
    // three neighbouring diagonals at a time: a_d + 256 a_(d+1) + 65536 a_(d+2) is an exact double (|a| < 2^31: below 2^47.1), so a row costs two modular
    // products (by 2^24 and 2^48 mod q_j) instead of one per diagonal
    const double c24 = BzF::center(16777216.0, mq), c48 = BzF::center(__dmul_rn(c24, 16777216.0), mq);       // exact: |c24| <= 2^24
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const uint32_t row = (uint32_t)(r & 3) + 8u * (uint32_t)(r >> 2) + 4u * half, o = g * M + mt * 32 + row;
        if (mt * 32 + row >= M) continue;
        auto group = [&](int d0) {
            double t = (double)acc[d0][r];
            if (d0 + 1 < D) t = __fma_rn((double)acc[d0 + 1 < D ? d0 + 1 : d0][r], 256.0, t);
            if (d0 + 2 < D) t = __fma_rn((double)acc[d0 + 2 < D ? d0 + 2 : d0][r], 65536.0, t);
            return t;
        };
        double v = __dadd_rn(group(0), BzF::mulmod(group(3), c24, mq));
        if (D > 6) v = __dadd_rn(v, BzF::mulmod(group(6), c48, mq));
        uint64_t res = BzF::to_u64(v, mq);
        if constexpr (ABS) {
            if (!out_idx[o]) continue;
            if (bias_idx && bias_idx[o] && limb < k) { const uint64_t bv = gmem(bias_idx[o])[(size_t)ctile * 32 + col]; if (bv) res = addmod(res, scale_plain(C, bv, j), qm.q); }
            gmem_w(out_idx[o])[e] = res;
        } else {
            if (out_idx[o] < 0) continue;
            if (bias && limb < k) { const uint64_t bv = bias[(size_t)bias_idx[o] * bu + (size_t)ctile * 32 + col]; if (bv) res = addmod(res, scale_plain(C, bv, j), qm.q); }
            out[(size_t)(obase + (uint32_t)out_idx[o]) * ou + e] = res;
        }
    }
}

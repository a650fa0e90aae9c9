// Launchers of the key-switch kernels for ONE arithmetic policy (KS_POLICY, KS_NAME): included by cn_l_ks_u64.hip, cn_l_ks_f64.hip
// and cn_l_ks_f64l.hip, so that the three policies compile in parallel.
#include "cn_runtime.h"
#include "cn_k_ks.hip.h"

typedef KS_POLICY AR;
static constexpr bool kF64 = std::is_same<typename AR::T, double>::value;

static constexpr size_t PAIR14_LDS = (size_t)ntt_lds_words(8192) * 8 + 65536;      // k_keyswitch_pair14: exchange image + 64 KiB (the parked next digit / staging window)
template <class K> static int big_lds(K kern, size_t bytes) {
    HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
// dynamic LDS of the fused key switch with LDS-resident forward twiddles: exchange image + table
template <int L> static size_t ks_twl_lds() { return ((size_t)ntt_lds_words(1u << L) + (1u << L)) * 8; }
template <int L> static int set_attrs_l(size_t bytes) {
    CHECK(big_lds(k_keyswitch_rr<L, AR>, bytes)); CHECK(big_lds(k_keyswitch_rr<L, AR, 1, false, true>, bytes));
    if constexpr (KsFwd<AR, L>::lds) { CHECK(big_lds(k_keyswitch_rr<L, AR, 1, true>, ks_twl_lds<L>())); CHECK(big_lds(k_keyswitch_rr<L, AR, 1, true, true>, ks_twl_lds<L>())); }
    CHECK(big_lds(k_ks_digit_mac<L, AR>, bytes)); CHECK(big_lds(k_ks_limb_mac<L, AR>, bytes)); CHECK(big_lds(k_ks_sum_intt<L, AR>, bytes));
    return 0;
}
static int set_attrs(uint32_t logn, size_t bytes) {
    if constexpr (kF64) { if (logn == 12) { CHECK(big_lds(k_keyswitch_rr<12, AR, 1, true>, ks_twl_lds<12>())); CHECK(big_lds(k_keyswitch_rr<12, AR, 1, true, true>, ks_twl_lds<12>())); } }   // N = 4096: image + LDS twiddle table = 66.5 KiB
    if (logn == 13) { CHECK(set_attrs_l<13>(bytes)); CHECK(big_lds(k_keyswitch_rr<13, AR, 4>, bytes)); }
    if (logn == 14) {
        CHECK(set_attrs_l<14>(bytes));
        if constexpr (kF64) {
            CHECK(big_lds(k_keyswitch_split14<AR>, (size_t)ntt_lds_words(8192) * 8)); CHECK(big_lds(k_keyswitch_split14<AR, true>, (size_t)ntt_lds_words(8192) * 8));
            CHECK(big_lds(k_keyswitch_pair14<AR, false, false>, PAIR14_LDS)); CHECK(big_lds(k_keyswitch_pair14<AR, true, false>, PAIR14_LDS));
            CHECK(big_lds(k_keyswitch_pair14<AR, false, true>, PAIR14_LDS)); CHECK(big_lds(k_keyswitch_pair14<AR, true, true>, PAIR14_LDS));
        }
    }
    return 0;
}
// the LDS copy of the twiddle table pays once a workgroup runs enough digit transforms of its modulus
static const uint32_t KS_TWL_MIN_DIGITS = 12;
template <int L, int MINW, bool XI> static void launch_fused_x(cn_ctx *c, const KsArgs &a) {
    const uint32_t tot = a.galois ? c->hc.gk_tot : c->hc.rl_tot;
    if constexpr (KsFwd<AR, L>::lds && MINW == 1) {
        if (tot >= KS_TWL_MIN_DIGITS) {
            hipLaunchKernelGGL((k_keyswitch_rr<L, AR, 1, true, XI>), dim3(a.cnt * c->hc.k), dim3(NttPlan<L>::NT), ks_twl_lds<L>(), c->stream, a.target, a.tstride, a.add0, a.add1,
                               a.astride, (const void *)a.key, a.out, c->dc, a.galois, a.accmax, a.extra, a.xstride, a.out_tab, a.xcd_cts);
            return;
        }
    }
    hipLaunchKernelGGL((k_keyswitch_rr<L, AR, MINW, false, XI>), dim3(a.cnt * c->hc.k), dim3(NttPlan<L>::NT), (size_t)ntt_lds_words(1u << L) * 8, c->stream, a.target, a.tstride,
                       a.add0, a.add1, a.astride, (const void *)a.key, a.out, c->dc, a.galois, a.accmax, a.extra, a.xstride, a.out_tab, a.xcd_cts);
}
template <int L, int MINW = 1> static void launch_fused(cn_ctx *c, const KsArgs &a) {
    if constexpr (MINW == 1) { if (c->hc.ks_xi) { launch_fused_x<L, 1, true>(c, a); return; } }      // (the 128-VGPR A/B variant exists for the default convention only)
    launch_fused_x<L, MINW, false>(c, a);
}
template <int L> static void launch_two_phase(cn_ctx *c, const KsArgs &a) {
    const uint32_t tot = a.galois ? c->hc.gk_tot : c->hc.rl_tot, k = c->hc.k;
    const size_t lds = (size_t)ntt_lds_words(1u << L) * 8;
    if (a.mode == 2) {              // one partial per (ct, source limb): k*k workgroups per ciphertext, k partials to sum
        hipLaunchKernelGGL((k_ks_limb_mac<L, AR>), dim3(a.cnt * k * k), dim3(NttPlan<L>::NT), lds, c->stream, a.target, a.tstride, (const void *)a.key, c->ks_part,
                           c->dc, a.galois, a.accmax, a.perm_elt, a.items);
        hipLaunchKernelGGL((k_ks_sum_intt<L, AR>), dim3(a.cnt * k * 2), dim3(NttPlan<L>::NT), lds, c->stream, (const void *)c->ks_part, a.add0, a.add1, a.astride,
                           a.out, c->dc, k, 0xffffffffu, a.extra, a.xstride, a.out_tab, a.perm_elt, a.items);
    } else {                        // one partial per (ct, digit)
        hipLaunchKernelGGL((k_ks_digit_mac<L, AR>), dim3(a.cnt * tot * k), dim3(NttPlan<L>::NT), lds, c->stream, a.target, a.tstride, (const void *)a.key, c->ks_part,
                           c->dc, a.galois, tot, a.perm_elt, a.items);
        hipLaunchKernelGGL((k_ks_sum_intt<L, AR>), dim3(a.cnt * k * 2), dim3(NttPlan<L>::NT), lds, c->stream, (const void *)c->ks_part, a.add0, a.add1, a.astride,
                           a.out, c->dc, tot, a.accmax, a.extra, a.xstride, a.out_tab, a.perm_elt, a.items);
    }
    cn_launch_count(c);
}
template <int L> static void launch_rr(cn_ctx *c, const KsArgs &a) {
    if (a.mode) { launch_two_phase<L>(c, a); return; }
    if constexpr (L == 13) { if (c->ks_tight && !c->hc.ks_xi) { launch_fused<L, 4>(c, a); return; } }      // 128-VGPR variant (A/B only)
    launch_fused<L>(c, a);
}
static bool launch(cn_ctx *c, const KsArgs &a) {
    switch (c->hc.logn) {
        case 10: launch_rr<10>(c, a); return true;
        case 11: launch_rr<11>(c, a); return true;
        case 12: launch_rr<12>(c, a); return true;
        case 13: launch_rr<13>(c, a); return true;
        case 14: launch_rr<14>(c, a); return true;
        default: return false;
    }
}
// N = 16384 as two 8192-point halves per limb: partial halves into c->ks_part (the caller runs k_ks_combine14 behind it)
static bool split14(cn_ctx *c, const KsArgs &a) {
    if constexpr (kF64) {
        if (c->hc.ks_xi)
            hipLaunchKernelGGL((k_keyswitch_split14<AR, true>), dim3(a.cnt * c->hc.k * 2), dim3(NttPlan<13>::NT), (size_t)ntt_lds_words(8192) * 8, c->stream, a.target, a.tstride,
                               (const void *)a.key, (uint64_t *)c->ks_part, c->dc, a.galois, a.accmax);
        else
            hipLaunchKernelGGL((k_keyswitch_split14<AR>), dim3(a.cnt * c->hc.k * 2), dim3(NttPlan<13>::NT), (size_t)ntt_lds_words(8192) * 8, c->stream, a.target, a.tstride,
                               (const void *)a.key, (uint64_t *)c->ks_part, c->dc, a.galois, a.accmax);
        return true;
    }
    return false;
}
// N = 16384 in one launch: both halves per (ciphertext, output limb) workgroup (k_keyswitch_pair14); a.target = sigma(c1) for rotations
template <bool XI, bool WHOLE> static void launch_pair14(cn_ctx *c, const KsArgs &a) {
    if constexpr (kF64)
        hipLaunchKernelGGL((k_keyswitch_pair14<AR, XI, WHOLE>), dim3(a.cnt * c->hc.k), dim3(NttPlan<13>::NT), PAIR14_LDS, c->stream, a.target, a.tstride, a.add0, a.add1, a.astride,
                           (const void *)a.key, a.out, (double *)c->ks_part, c->dc, a.galois, a.accmax, a.extra, a.xstride, a.out_tab, a.perm_elt, a.next_elt, a.next_out, a.xcd_cts);
}
static bool pair14(cn_ctx *c, const KsArgs &a0) {
    if constexpr (kF64) {
        // lazy accumulators of THIS kernel: the forward half-transform (L = 13: five stages behind its one recentring) leaves |v| <= 4.82 q, a term
        // mulmod(v, key) is then at most (1/2 + 0.1875 x 4.82) q = 1.41 q (cn_ntt_core.hip.h), and 2^53 / q >= 16 for q < 2^49: 8 terms (+ the q/2 a recentred
        // accumulator starts from) stay below 11.8 q.  The generic bound of do_keyswitch (2.1 q per term, sum below 2^52) allows 2 at 49 bits: a recentring of
        // both accumulator sets every other digit, 2 % of the kernel's FP64 instructions at the reference's N = 16384 parameters.
        KsArgs a = a0;
        {
            uint64_t qm = 0; for (uint32_t j = 0; j < c->hc.k; j++) qm = std::max(qm, c->hc.q[j].q);
            const int bits = 64 - __builtin_clzll(qm);
            if (bits >= 45 && bits <= 49) a.accmax = std::max(a.accmax, 1u << (52 - bits));
        }
        // one digit per source limb that covers the limb (the reference's N = 16384 parameter sets, dbc 60): the digit is the word itself
        const int dbc = a.galois ? c->hc.gdbc : c->hc.dbc;
        uint64_t qmax = 0; for (uint32_t j = 0; j < c->hc.k; j++) qmax = std::max(qmax, c->hc.q[j].q);
        const bool whole = (a.galois ? c->hc.gk_tot : c->hc.rl_tot) == c->hc.k && dbc < 64 && (qmax >> dbc) == 0;
        if (c->hc.ks_xi) { if (whole) launch_pair14<true, true>(c, a); else launch_pair14<true, false>(c, a); }
        else { if (whole) launch_pair14<false, true>(c, a); else launch_pair14<false, false>(c, a); }
        return true;
    }
    return false;
}
#ifndef __HIP_DEVICE_COMPILE__      // host-side table (in the device pass a const global would be emitted as device data)
extern const KsOps KS_NAME = {set_attrs, launch, split14, pair14};
#endif

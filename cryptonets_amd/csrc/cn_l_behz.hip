// Launchers of the element-wise BEHZ steps (base extension q -> Bsk u {m~}; x t, fast floor, Shenoy-Kumaresan back to q), one
// instantiation per coefficient-modulus count.
#include "cn_runtime.h"
#include "cn_k_behz.hip.h"

// NB = primes of the auxiliary base B (the auxiliary limbs are B then m_sk): k (SEAL's shape, and the small base where it is large enough)
// or k + 1 (N = 16384 with 7 - 8 data primes of 48-49 bits: one more small prime makes the Shenoy-Kumaresan bound hold)
template <int K, int NB> static void launch_extend(cn_ctx *c, const uint64_t *src, uint32_t stride, const uint64_t *const *tab, uint64_t *aq, uint64_t *ab, uint32_t cnt) {
    if (c->hc.behz_f64 && c->use_f64) hipLaunchKernelGGL((k_behz_extend_f64<K, NB>), dim3(cnt * 2 * c->chunks), dim3(c->bs), 0, c->stream, src, stride, tab, aq, ab, c->dc, c->chunks);
    else hipLaunchKernelGGL((k_behz_extend<K, NB>), dim3(cnt * 2 * c->chunks), dim3(c->bs), 0, c->stream, src, stride, tab, aq, ab, c->dc, c->chunks);
}
template <int K, int NB> static void launch_floor(cn_ctx *c, const uint64_t *dq, const uint64_t *db, uint64_t *out, uint32_t cnt) {
    if (c->hc.behz_f64 && c->use_f64) hipLaunchKernelGGL((k_behz_floor_f64<K, NB>), dim3(cnt * 3 * c->chunks), dim3(c->bs), 0, c->stream, dq, db, out, c->dc, c->chunks);
    else hipLaunchKernelGGL((k_behz_floor<K, NB>), dim3(cnt * 3 * c->chunks), dim3(c->bs), 0, c->stream, dq, db, out, c->dc, c->chunks);
}
#define DISPATCH_K(fn, ...) do { \
    if (c->hc.kb == c->hc.k + 1) switch (c->hc.k) { \
        case 1: fn<1, 1>(__VA_ARGS__); break; case 2: fn<2, 2>(__VA_ARGS__); break; case 3: fn<3, 3>(__VA_ARGS__); break; \
        case 4: fn<4, 4>(__VA_ARGS__); break; case 5: fn<5, 5>(__VA_ARGS__); break; case 6: fn<6, 6>(__VA_ARGS__); break; \
        case 7: fn<7, 7>(__VA_ARGS__); break; case 8: fn<8, 8>(__VA_ARGS__); break; case 9: fn<9, 9>(__VA_ARGS__); break; \
        default: return cn_fail(CN_ERR_ARG, "ciphertext multiply supports at most 9 coefficient moduli"); } \
    else if (c->hc.kb == c->hc.k + 2) switch (c->hc.k) { \
        case 6: fn<6, 7>(__VA_ARGS__); break; case 7: fn<7, 8>(__VA_ARGS__); break; case 8: fn<8, 9>(__VA_ARGS__); break; case 9: fn<9, 10>(__VA_ARGS__); break; \
        default: return cn_fail(CN_ERR_ARG, "internal: auxiliary base of k + 1 primes for k = %u", c->hc.k); } \
    else return cn_fail(CN_ERR_ARG, "internal: auxiliary base size"); } while (0)

int cn_l_behz_extend(cn_ctx *c, const uint64_t *src, uint32_t stride, const uint64_t *const *src_tab, uint64_t *aq, uint64_t *ab, uint32_t cnt) {
    DISPATCH_K(launch_extend, c, src, stride, src_tab, aq, ab, cnt);
    HIPCHK(hipGetLastError()); cn_launch_count(c);
    return 0;
}
int cn_l_behz_floor(cn_ctx *c, const uint64_t *dq, const uint64_t *db, uint64_t *out, uint32_t cnt) {
    DISPATCH_K(launch_floor, c, dq, db, out, cnt);
    HIPCHK(hipGetLastError()); cn_launch_count(c);
    return 0;
}

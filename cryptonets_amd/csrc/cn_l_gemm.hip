// Launchers of the scalar GEMM kernels (HOT LOOP A).  One translation unit per kernel family: hipcc compiles them in parallel.
#include "cn_runtime.h"
#include <algorithm>
#include "cn_k_gemm.hip.h"

// order 1 (slice-major) needs the slices in multiples of 8 and is launched as a 2-D grid: x = (XCD, output tile, group), y = slice / 8 (gemm_block_coords)
struct GemmGrid { dim3 grid; uint32_t order; };
static GemmGrid gemm_grid(cn_ctx *c, const GemmLaunch &g, uint32_t mtiles) {
    const uint32_t slices = c->chunks * g.polys * c->hc.k;
    if (g.order == 1 && (slices & 7) == 0 && (c->chunks & (c->chunks - 1)) == 0 && (uint64_t)8 * mtiles * g.G < (1ull << 31) && slices / 8 < 65536)
        return {dim3(8 * mtiles * g.G, slices / 8), 1u};
    return {dim3((uint32_t)((size_t)slices * mtiles * g.G)), 0u};
}
template <int MT, bool ABS> static void launch_int(cn_ctx *c, const GemmLaunch &g) {
    const uint32_t mtiles = (g.M + MT - 1) / MT;
    const GemmGrid gg = gemm_grid(c, g, mtiles);
    hipLaunchKernelGGL((k_scalar_gemm<MT, ABS>), gg.grid, dim3(c->bs), 0, c->stream, g.in, g.idx, (const uint64_t *)g.W, g.oidx, g.bias, g.bidx, g.out, c->dc,
                       c->chunks, g.G, g.M, g.K, mtiles, g.lazy, g.Kp, g.obase, g.polys, gg.order, GemmUnits{g.in_unit, g.out_unit, g.bias_unit});
}
template <int MT, bool ABS> static void launch_f64(cn_ctx *c, const GemmLaunch &g) {
    const uint32_t mtiles = (g.M + MT - 1) / MT;
    const GemmGrid gg = gemm_grid(c, g, mtiles);
    if (g.one) hipLaunchKernelGGL((k_scalar_gemm_f64<MT, 1, 0, ABS>), gg.grid, dim3(c->bs), 0, c->stream, g.in, g.idx, (const double *)g.W, g.oidx, g.bias,
                                  g.bidx, g.out, c->dc, c->chunks, g.G, g.M, g.K, mtiles, g.lazy, g.Kp, g.obase, g.polys, gg.order, gemm_f64_rows(g.K), GemmUnits{g.in_unit, g.out_unit, g.bias_unit});
    else if (g.two) hipLaunchKernelGGL((k_scalar_gemm_f64<MT, 2, 22, ABS>), gg.grid, dim3(c->bs), 0, c->stream, g.in, g.idx, (const double *)g.W, g.oidx, g.bias,
                                  g.bidx, g.out, c->dc, c->chunks, g.G, g.M, g.K, mtiles, g.lazy, g.Kp, g.obase, g.polys, gg.order, gemm_f64_rows(g.K), GemmUnits{g.in_unit, g.out_unit, g.bias_unit});
    else hipLaunchKernelGGL((k_scalar_gemm_f64<MT, 3, 17, ABS>), gg.grid, dim3(c->bs), 0, c->stream, g.in, g.idx, (const double *)g.W, g.oidx, g.bias,
                            g.bidx, g.out, c->dc, c->chunks, g.G, g.M, g.K, mtiles, g.lazy, g.Kp, g.obase, g.polys, gg.order, gemm_f64_rows(g.K), GemmUnits{g.in_unit, g.out_unit, g.bias_unit});
}
template <bool ABS> static int launch(cn_ctx *c, const GemmLaunch &g) {
    if (g.small) {
        switch (g.MT) {
            case 20: launch_f64<20, ABS>(c, g); break;
            case 10: launch_f64<10, ABS>(c, g); break;
            case 5: launch_f64<5, ABS>(c, g); break;
            case 1: launch_f64<1, ABS>(c, g); break;
            default: return cn_fail(CN_ERR_ARG, "internal: scalar GEMM tile %u", g.MT);
        }
    } else {
        switch (g.MT) {
            case 10: launch_int<10, ABS>(c, g); break;
            case 5: launch_int<5, ABS>(c, g); break;
            case 1: launch_int<1, ABS>(c, g); break;
            default: return cn_fail(CN_ERR_ARG, "internal: scalar GEMM tile %u", g.MT);
        }
    }
    HIPCHK(hipGetLastError());
    cn_launch_count(c);
    return 0;
}
int cn_l_gemm(cn_ctx *c, const GemmLaunch &g) { return g.abs ? launch<true>(c, g) : launch<false>(c, g); }

template <int P, bool ABS> static void launch_mfma(cn_ctx *c, const GemmLaunch &g) {
    const uint32_t mgroups = (g.mtiles + 3) / 4;
    const size_t blocks = (size_t)g.G * mgroups * g.polys * c->hc.k * (c->hc.n / 32);
    hipLaunchKernelGGL((k_scalar_gemm_mfma<P, ABS>), dim3((uint32_t)blocks), dim3(256), 0, c->stream, g.in, g.idx, (const int8_t *)g.W, g.oidx, g.bias, g.bidx, g.out,
                       c->dc, g.G, g.M, g.mtiles, g.ksteps, g.obase, g.polys, GemmUnits{g.in_unit, g.out_unit, g.bias_unit});
}
template <bool ABS> static int launch_mfma_p(cn_ctx *c, const GemmLaunch &g) {
    switch (g.P) {
        case 1: launch_mfma<1, ABS>(c, g); break;
        case 2: launch_mfma<2, ABS>(c, g); break;
        case 3: launch_mfma<3, ABS>(c, g); break;
        default: return cn_fail(CN_ERR_ARG, "internal: %u weight digit planes", g.P);
    }
    HIPCHK(hipGetLastError());
    cn_launch_count(c);
    return 0;
}
int cn_l_gemm_mfma(cn_ctx *c, const GemmLaunch &g) { return g.abs ? launch_mfma_p<true>(c, g) : launch_mfma_p<false>(c, g); }

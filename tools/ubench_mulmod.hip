// How much instruction-level parallelism does the exact-FP64 modular multiply need?  Each thread runs ILP independent chains of
//   p = a*w; e = fma(a,w,-p); h = rint(p*qinv); t = fma(-h,q,p); a = t + e        (6 dependent FP64 instructions)
// for ITERS iterations, with WAVES waves per SIMD resident (launch geometry + dynamic LDS to pin the occupancy).
// Prints FP64 wave-instructions per SIMD-cycle at the measured rate (GPU time from events, clock from the v_fma_f64 peak run).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
template <int ILP>
__global__ void __launch_bounds__(256) k_chain(double *out, double w, double q, double qinv) {
    double a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) a[i] = (double)(threadIdx.x * 131 + i * 7 + 1);
    for (int it = 0; it < ITERS; it++) {
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
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP> void run(int waves_per_simd, double *buf) {
    // 256 threads = 4 waves = 1 wave per SIMD per block; LDS request limits the blocks per CU to `waves_per_simd`
    const int blocks = 256 * waves_per_simd;
    const size_t lds = waves_per_simd >= 8 ? 0 : (size_t)(160 * 1024 / waves_per_simd) - 1024;
    hipFuncSetAttribute((const void *)k_chain<ILP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double q = 8796092792833.0, w = 1234567891011.0;
    hipLaunchKernelGGL(k_chain<ILP>, dim3(blocks), dim3(256), lds, 0, buf, w, q, 1.0 / q);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_chain<ILP>, dim3(blocks), dim3(256), lds, 0, buf, w, q, 1.0 / q); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double instr_per_simd = (double)waves_per_simd * ITERS * ILP * 6;      // every SIMD runs waves_per_simd waves
    printf("waves/SIMD %d  ILP %2d : %7.3f ms  %6.2f ns per FP64 instr per SIMD  (%.2f cycles at 2.1 GHz)\n", waves_per_simd, ILP, best,
           best * 1e6 / instr_per_simd, best * 1e6 / instr_per_simd * 2.1);
}
int main() {
    double *buf; hipMalloc(&buf, 256 * 8 * 256 * 8);
    for (int w : {1, 2, 4, 8}) { run<1>(w, buf); run<2>(w, buf); run<4>(w, buf); run<8>(w, buf); run<16>(w, buf); }
    return 0;
}

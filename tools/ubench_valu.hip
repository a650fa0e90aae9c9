// Micro-benchmark of the VALU instructions that bound 64-bit modular arithmetic on gfx950:
// v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32 (integer), v_fma_f64 / v_mul_f64 (FP64 path for <=50-bit primes),
// v_add_co_u32 / v_lshl_add_u64 (carry chains).  Prints wave-instructions per ns for the whole chip and the implied
// cycles per wave-instruction per SIMD at 2.4 GHz.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define ITERS 4096
#define CHAINS 8
#define OPLOOP(NAME, DECL, BODY, SINK)                                                         \
    __global__ void __launch_bounds__(256) NAME(uint64_t *out, uint32_t seed) {                 \
        DECL;                                                                                    \
        for (int it = 0; it < ITERS; it++) {                                                     \
            BODY                                                                                 \
        }                                                                                        \
        SINK;                                                                                    \
    }
// 32-bit chains
#define D32 uint32_t a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = seed | 1
#define S32 out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7
#define B32(OP) asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n" \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
OPLOOP(k_mul_lo, D32, B32("v_mul_lo_u32"), S32)
OPLOOP(k_mul_hi, D32, B32("v_mul_hi_u32"), S32)
OPLOOP(k_add_u32, D32, B32("v_add_u32"), S32)
// 64-bit chains
#define D64 uint64_t a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; uint32_t b = seed | 1, c = seed + 77
#define S64 out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7
#define BMAD asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\nv_mad_u64_u32 %1, vcc, %8, %9, %1\nv_mad_u64_u32 %2, vcc, %8, %9, %2\nv_mad_u64_u32 %3, vcc, %8, %9, %3\n" \
                          "v_mad_u64_u32 %4, vcc, %8, %9, %4\nv_mad_u64_u32 %5, vcc, %8, %9, %5\nv_mad_u64_u32 %6, vcc, %8, %9, %6\nv_mad_u64_u32 %7, vcc, %8, %9, %7\n" \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
OPLOOP(k_mad_u64_u32, D64, BMAD, S64)
#define BLSHLADD asm volatile("v_lshl_add_u64 %0, %0, 0, %8\nv_lshl_add_u64 %1, %1, 0, %8\nv_lshl_add_u64 %2, %2, 0, %8\nv_lshl_add_u64 %3, %3, 0, %8\n" \
                              "v_lshl_add_u64 %4, %4, 0, %8\nv_lshl_add_u64 %5, %5, 0, %8\nv_lshl_add_u64 %6, %6, 0, %8\nv_lshl_add_u64 %7, %7, 0, %8\n" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a7 | 1));
OPLOOP(k_add_u64, D64, BLSHLADD, S64)
// FP64 chains
#define DF64 double a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0000001, c = 1e-9
#define SF64 out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define BFMA asm volatile("v_fma_f64 %0, %0, %8, %9\nv_fma_f64 %1, %1, %8, %9\nv_fma_f64 %2, %2, %8, %9\nv_fma_f64 %3, %3, %8, %9\n" \
                          "v_fma_f64 %4, %4, %8, %9\nv_fma_f64 %5, %5, %8, %9\nv_fma_f64 %6, %6, %8, %9\nv_fma_f64 %7, %7, %8, %9\n" \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
OPLOOP(k_fma_f64, DF64, BFMA, SF64)
#define BMULF asm volatile("v_mul_f64 %0, %0, %8\nv_mul_f64 %1, %1, %8\nv_mul_f64 %2, %2, %8\nv_mul_f64 %3, %3, %8\n" \
                           "v_mul_f64 %4, %4, %8\nv_mul_f64 %5, %5, %8\nv_mul_f64 %6, %6, %8\nv_mul_f64 %7, %7, %8\n" \
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
OPLOOP(k_mul_f64, DF64, BMULF, SF64)
#define BADDF asm volatile("v_add_f64 %0, %0, %8\nv_add_f64 %1, %1, %8\nv_add_f64 %2, %2, %8\nv_add_f64 %3, %3, %8\n" \
                           "v_add_f64 %4, %4, %8\nv_add_f64 %5, %5, %8\nv_add_f64 %6, %6, %8\nv_add_f64 %7, %7, %8\n" \
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
OPLOOP(k_add_f64, DF64, BADDF, SF64)
#define BRNDF asm volatile("v_rndne_f64 %0, %0\nv_rndne_f64 %1, %1\nv_rndne_f64 %2, %2\nv_rndne_f64 %3, %3\nv_rndne_f64 %4, %4\nv_rndne_f64 %5, %5\nv_rndne_f64 %6, %6\nv_rndne_f64 %7, %7\n" \
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
OPLOOP(k_rndne_f64, DF64, BRNDF, SF64)
#define BFMA32 asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n" \
                            "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n" \
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define DF32 float a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0000001f, c = 1e-9f
OPLOOP(k_fma_f32, DF32, BFMA32, SF64)

template <class K> void run(const char *name, K kern, uint64_t *buf) {
    const int blocks = 256 * 8, threads = 256;      // 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, buf, 3u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, buf, 3u + r); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double wave_instr = (double)blocks * (threads / 64) * ITERS * CHAINS;
    double per_ns = wave_instr / (best * 1e6);
    double cyc = 2.4 * 1024.0 / per_ns;             // cycles per wave-instruction per SIMD at 2.4 GHz (1024 SIMDs)
    printf("%-16s %8.3f ms  %9.1f wave-instr/ns  ~%5.2f cyc/wave-instr/SIMD@2.4GHz  %7.2f T lane-ops/s\n", name, best, per_ns, cyc, per_ns * 64 / 1e3);
}
int main() {
    uint64_t *buf; hipMalloc(&buf, 256 * 8 * 256 * 8);
    run("v_add_u32", k_add_u32, buf);
    run("v_fma_f32", k_fma_f32, buf);
    run("v_mul_lo_u32", k_mul_lo, buf);
    run("v_mul_hi_u32", k_mul_hi, buf);
    run("v_mad_u64_u32", k_mad_u64_u32, buf);
    run("v_lshl_add_u64", k_add_u64, buf);
    run("v_fma_f64", k_fma_f64, buf);
    run("v_mul_f64", k_mul_f64, buf);
    run("v_add_f64", k_add_f64, buf);
    run("v_rndne_f64", k_rndne_f64, buf);
    return 0;
}

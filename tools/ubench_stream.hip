// HBM ceiling of the batched transform's ACCESS PATTERN without its arithmetic: what does the memory system deliver for
//   (a) a plain 16 B/lane copy (the guide's 6.3 TB/s figure),
//   (b) the transform's pattern in place: one 512-thread workgroup per 64 KiB limb, 16 loads of 8 B per lane at a 4 KiB stride, 8 stores of
//       16 B per lane, same addresses (k_ntt_rr reads a limb and writes it back),
//   (c) the same out of place,
//   (d) (b) with ~5 us of dependent FP64 work between the loads and the stores (the transform's duration), two workgroups per CU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_stream.hip -o tools/ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr uint32_t N = 8192, NT = 512;
__global__ void __launch_bounds__(256) k_copy(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
template <int WORK>
__global__ void __launch_bounds__(NT, 4) k_limb(const uint64_t *__restrict__ src, uint64_t *dst, double w, double q, double qinv) {
    extern __shared__ double lds[];                      // sized like the exchange image: two workgroups per CU
    const uint64_t *x = src + (size_t)blockIdx.x * N;
    uint64_t *y = dst + (size_t)blockIdx.x * N;
    uint64_t v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = x[r * NT + threadIdx.x];
    if (WORK) {
        double a[16];
#pragma unroll
        for (int r = 0; r < 16; r++) a[r] = (double)(uint32_t)v[r];
        for (int it = 0; it < WORK; it++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const double p = a[r] * w, e = __fma_rn(a[r], w, -p), h = __builtin_rint(p * qinv);
                a[r] = __fma_rn(-h, q, p) + e;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] += (uint64_t)(int64_t)a[r];
        if (a[0] == 1.2345e-300) lds[threadIdx.x] = a[1];
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        ulonglong2 o; o.x = v[r] + 1; o.y = v[r + 1] + 1;
        *reinterpret_cast<ulonglong2 *>(y + (r / 2) * 2 * NT + threadIdx.x * 2) = o;
    }
}
template <class F> static float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best;
}
int main() {
    for (uint32_t limbs : {8450u, 40960u}) {
        const size_t bytes = (size_t)limbs * N * 8;
        uint64_t *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
        const size_t lds = 66 * 1024;
        hipFuncSetAttribute((const void *)k_limb<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void *)k_limb<14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const double q = 8796092792833.0;
        float t;
        t = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16); });
        printf("%6u limbs | (a) 16 B/lane copy                         : %8.1f us  %7.1f GB/s\n", limbs, t * 1e3, 2.0 * bytes / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_limb<0>, dim3(limbs), dim3(NT), lds, 0, a, a, 3.0, q, 1.0 / q); });
        printf("%6u limbs | (b) limb pattern, in place, no arithmetic   : %8.1f us  %7.1f GB/s\n", limbs, t * 1e3, 2.0 * bytes / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_limb<0>, dim3(limbs), dim3(NT), lds, 0, a, b, 3.0, q, 1.0 / q); });
        printf("%6u limbs | (c) limb pattern, out of place              : %8.1f us  %7.1f GB/s\n", limbs, t * 1e3, 2.0 * bytes / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_limb<14>, dim3(limbs), dim3(NT), lds, 0, a, a, 3.0, q, 1.0 / q); });
        printf("%6u limbs | (d) in place + 14 x 16 modular multiplies   : %8.1f us  %7.1f GB/s\n", limbs, t * 1e3, 2.0 * bytes / t / 1e6);
        hipFree(a); hipFree(b);
    }
    return 0;
}

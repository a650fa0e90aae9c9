// Where does the time of the register-radix transform go?  The forward N = 8192 FP64 transform of k_ntt_rr with pieces switched off
// (results are then meaningless - timing only): global loads / stores, LDS exchanges, butterfly arithmetic, the workgroup barrier;
// and the full kernel at 1 and 2 workgroups per CU.   hipcc -O3 --offload-arch=gfx950 -I cryptonets_amd/csrc tools/ubench_ntt.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "cn_ntt_core.hip.h"
enum { F_LOAD = 1, F_STORE = 2, F_LDS = 4, F_MATH = 8, F_BAR = 16, F_NT = 32, F_NTS = 64 };      // F_NT / F_NTS: non-temporal loads / stores
typedef ArF64T<0> AR;
constexpr int L = 13;
template <int F>
__global__ void __launch_bounds__(512) k_probe(uint64_t *data, const double *tw_, double q, double qinv) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *s = reinterpret_cast<double *>(smem);
    constexpr uint32_t n = 1u << L;
    constexpr int SA = NttPlan<L>::SA;
    const uint32_t tid = threadIdx.x;
    const AR::Mod m = {q, qinv};
    const AR::Tw tw = {(const NTT_GLOBAL double *)tw_};
    uint64_t *x = data + (size_t)blockIdx.x * n;
    double v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = (F & F_LOAD) ? AR::from_u64((F & F_NT) ? __builtin_nontemporal_load(x + pass_index<L, SA, 0>(tid, r)) : x[pass_index<L, SA, 0>(tid, r)]) : (double)(tid * 16 + r + blockIdx.x);
    if (F & F_MATH) fwd_stages<AR, L, SA, 0>(v, tw, m, tid);
    if (F & F_LDS) lds_put<double, L, SA, 0>(v, s, tid);
    if (F & F_BAR) __syncthreads();
    if (F & F_LDS) lds_get<double, L, 4, SA>(v, s, tid);
    if (F & F_MATH) fwd_stages<AR, L, 4, SA>(v, tw, m, tid);
    if (F & F_LDS) { lds_put<double, L, 4, SA>(v, s, tid); ntt_wave_sync(); lds_get<double, L, 4, SA + 4>(v, s, tid); }
    if (F & F_MATH) fwd_stages<AR, L, 4, SA + 4>(v, tw, m, tid);
    if (F & F_LDS) { lds_put<double, L, 4, SA + 4>(v, s, tid); ntt_wave_sync(); lds_get_tail<double, L>(v, s, tid); }
    if (F & F_MATH) fwd_tail<AR, L>(v, tw, m, tid);
    if (F & F_STORE) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            ulonglong2 o; o.x = AR::to_u64(v[r], m); o.y = AR::to_u64(v[r + 1], m);
            typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
            if (F & F_NTS) { ull2 ov; ov.x = o.x; ov.y = o.y; __builtin_nontemporal_store(ov, reinterpret_cast<ull2 *>(x + tail_index<L>(tid, r))); }
            else *reinterpret_cast<ulonglong2 *>(x + tail_index<L>(tid, r)) = o;
        }
    } else {
        double acc = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) acc += v[r];
        if (acc == 1.2345e300) x[tid] = 1;          // keeps the work alive without traffic
    }
}
template <int F> void run(const char *what, uint64_t *data, const double *tw, int limbs, size_t lds) {
    hipFuncSetAttribute((const void *)k_probe<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const double q = 8796092792833.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<F>, dim3(limbs), dim3(512), lds, 0, data, tw, q, 1.0 / q);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 6; r++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_probe<F>, dim3(limbs), dim3(512), lds, 0, data, tw, q, 1.0 / q); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-46s lds/wg %6zu B : %7.3f ms  (%.0f GB/s algorithmic)\n", what, lds, best, limbs * 8192.0 * 16 / (best * 1e6));
}
int main() {
    const int limbs = 8450;
    uint64_t *data; double *tw;
    hipMalloc(&data, (size_t)limbs * 8192 * 8); hipMalloc(&tw, 8192 * 8);
    std::vector<uint64_t> h((size_t)limbs * 8192);
    uint64_t st = 88172645463325252ull;
    for (auto &v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = st % 8796092792833ull; }
    hipMemcpy(data, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> t(8192);
    for (int i = 0; i < 8192; i++) t[i] = (double)(h[i] % 8796092792833ull);
    hipMemcpy(tw, t.data(), 8192 * 8, hipMemcpyHostToDevice);
    const size_t img = (size_t)ntt_lds_words(8192) * 8;
    run<31>("full", data, tw, limbs, img);
    run<31 + F_NT>("full, non-temporal loads", data, tw, limbs, img);
    run<31 + F_NTS>("full, non-temporal stores", data, tw, limbs, img);
    run<31 + F_NT + F_NTS>("full, non-temporal loads and stores", data, tw, limbs, img);
    run<3 + F_NT + F_NTS>("global load + store only, non-temporal", data, tw, limbs, img);
    run<3 + F_NTS>("global load + store only, non-temporal stores", data, tw, limbs, img);
    run<31>("full, 1 workgroup per CU", data, tw, limbs, 100 * 1024);
    run<15>("full without the barrier", data, tw, limbs, img);
    run<28>("no global traffic (math + LDS + barrier)", data, tw, limbs, img);
    run<8>("math only", data, tw, limbs, img);
    run<8>("math only, 1 workgroup per CU", data, tw, limbs, 100 * 1024);
    run<20>("LDS exchanges + barrier only", data, tw, limbs, img);
    run<3>("global load + store only", data, tw, limbs, img);
    run<1>("global load only", data, tw, limbs, img);
    run<23>("load + LDS + store (no math)", data, tw, limbs, img);
    run<11>("load + math + store (no LDS)", data, tw, limbs, img);
    return 0;
}

// Where does the time of the fused key switch go?  The loop structure of k_keyswitch_rr<13, ArF64T<0>, 1, true> (845 ciphertexts x 5
// output limbs, 25 digits each, LDS twiddle table, SGPR first-pass roots, 2 inverse transforms) with pieces switched off - timing only.
// (Measured and dropped from this file: 256-thread workgroups whose threads play two "virtual threads" of the 512-thread layout one
// after the other, so that two INDEPENDENT workgroups share a CU - 234 VGPRs, every load is followed by its own s_waitcnt: 5.7 ms
// against 3.0 ms.  The accumulators of 32 coefficients per thread do not leave the scheduler any registers.  Likewise the 512-thread
// layout with the accumulators parked in the output's place (read-modify-write through L2 per digit) and the source words re-read per
// digit, which fits 128 VGPRs and two workgroups per CU: 6.4 ms.  Differences below ~8 % between variants of this file are run-to-run
// noise (clock / power state): the same kernel measures 2.9-3.1 ms.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I cryptonets_amd/csrc tools/ubench_ks.hip -o tools/ubench_ks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "cn_ntt_core.hip.h"
enum { F_KEYS = 1, F_LDS = 2, F_MATH = 4, F_MAC = 8, F_BAR = 16, F_TWL = 32, F_RAW = 64, F_DB = 128, F_KLDS = 256, F_PRIO = 512 };   // F_KLDS: first key component of a digit prefetched into LDS with global_load_lds (no registers)   // F_DB: two LDS images, one barrier per digit
typedef ArF64T<0> AR;
constexpr int L = 13;
constexpr uint32_t N = 1u << L;
constexpr int SA = NttPlan<L>::SA;

template <int F, class FW> __device__ __forceinline__ void fwd_flagged(double (&v)[16], double *s, const typename FW::Tw &tw, const AR::Mod &m, uint32_t tid) {
    if (F & F_MATH) fwd_stages<FW, L, SA, 0>(v, tw, m, tid);
    if ((F & F_BAR) && !(F & F_DB)) __syncthreads();
    if (F & F_LDS) lds_put<double, L, SA, 0>(v, s, tid);
    if (F & F_BAR) __syncthreads();
    if (F & F_LDS) lds_get<double, L, 4, SA>(v, s, tid);
    if (F & F_MATH) fwd_stages<FW, L, 4, SA>(v, tw, m, tid);
    if (F & F_LDS) { lds_put<double, L, 4, SA>(v, s, tid); ntt_wave_sync(); lds_get<double, L, 4, SA + 4>(v, s, tid); }
    if (F & F_MATH) fwd_stages<FW, L, 4, SA + 4>(v, tw, m, tid);
    if (F & F_LDS) { lds_put<double, L, 4, SA + 4>(v, s, tid); ntt_wave_sync(); lds_get_tail<double, L>(v, s, tid); }
    if (F & F_MATH) fwd_tail<FW, L>(v, tw, m, tid);
}
template <int F>
__global__ void __launch_bounds__(512, 1) k_ks(const uint64_t *__restrict__ target, const double *__restrict__ key, uint64_t *out, const double *tw_, double q, double qinv) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *s = reinterpret_cast<double *>(smem);
    const uint32_t k = 5, tid = threadIdx.x, ct = blockIdx.x / k, j = blockIdx.x % k;
    if constexpr ((F & F_PRIO) != 0) { if (tid < 256) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }   // one wave of every SIMD runs ahead
    const AR::Mod m = {q, qinv};
    const size_t kn = (size_t)k * N;
    typedef typename std::conditional<(F & F_TWL) != 0, ArF64LdsT<0>, AR>::type FW0;
    typedef ArPassA<FW0> FW;
    typename FW::Tw fwt;
    const NTT_GLOBAL double *gtw = (const NTT_GLOBAL double *)tw_ + (size_t)j * 2 * N;
    if constexpr ((F & F_TWL) != 0) {
        double *tws = s + ntt_lds_words(N) * ((F & F_DB) ? 2 : 1);
        for (uint32_t i = tid * 2; i < N; i += 1024) { tws[i] = gtw[i]; tws[i + 1] = gtw[i + 1]; }
        fwt.w = (const __attribute__((address_space(3))) double *)tws;
        __syncthreads();
    } else fwt.w = gtw;
    ntt_load_pass_a<SA>(fwt, gtw);
    const AR::Tw ivt = {gtw + N};
    double acc0[16], acc1[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0; acc1[r] = 0; }
    const double *kp = key;
    for (uint32_t l = 0; l < k; l++) {
        const uint64_t *src = target + (size_t)ct * 3 * kn + 2 * kn + (size_t)l * N;
        uint64_t raw[16];
        {
            uint32_t t0 = tid;
            asm volatile("" : "+v"(t0));
#pragma unroll
            for (int r = 0; r < 16; r++) raw[r] = (F & F_RAW) ? src[pass_index<L, SA, 0>(t0, r)] : (uint64_t)(t0 * 977 + r * 131 + l) * 0x9E3779B97F4A7C15ull >> 20;
        }
        for (uint32_t d = 0; d < 5; d++, kp += 2 * kn) {
            const int sh = 10 * (int)d;
            uint32_t tl = tid;
            asm volatile("" : "+v"(tl));
            double v[16];
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = AR::from_u64((raw[r] >> sh) & 1023);
            const double *k0 = kp + (size_t)j * N, *k1 = kp + kn + (size_t)j * N;
            double *kbuf = s + ntt_lds_words(N) + ((tid >> 6) * 8) * 128;            // this wave's 8 KiB: [pair j][lane] 16 B
            if constexpr ((F & F_KLDS) != 0) {
#pragma unroll
                for (int jj = 0; jj < 8; jj++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(k0 + tail_index<L>(tl, 2 * jj)),
                                                     (__attribute__((address_space(3))) void *)(kbuf + jj * 128), 16, 0, 0);
            }
            fwd_flagged<F, FW>(v, (F & F_DB) ? s + ((l * 5 + d) & 1) * ntt_lds_words(N) : s, fwt, m, tl);
            if constexpr ((F & F_KLDS) != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const uint32_t pos = tail_index<L>(tl, r);
                struct alignas(16) P2 { double a, b; };
                P2 a, b;
                if (F & F_KLDS) { a = *reinterpret_cast<const P2 *>(kbuf + (r >> 1) * 128 + (tl & 63) * 2); b = *reinterpret_cast<const P2 *>(k1 + pos); }
                else if (F & F_KEYS) { a = *reinterpret_cast<const P2 *>(k0 + pos); b = *reinterpret_cast<const P2 *>(k1 + pos); }
                else { a = P2{q - 3.0 - r, 12345.0 + pos}; b = P2{q - 5.0 - r, 54321.0 + pos}; }
                if (F & F_MAC) {
                    acc0[r] = __dadd_rn(acc0[r], AR::mulmod(v[r], a.a, m)); acc0[r + 1] = __dadd_rn(acc0[r + 1], AR::mulmod(v[r + 1], a.b, m));
                    acc1[r] = __dadd_rn(acc1[r], AR::mulmod(v[r], b.a, m)); acc1[r + 1] = __dadd_rn(acc1[r + 1], AR::mulmod(v[r + 1], b.b, m));
                } else {
                    acc0[r] += v[r] + a.a; acc0[r + 1] += v[r + 1] + a.b; acc1[r] += v[r] + b.a; acc1[r + 1] += v[r + 1] + b.b;
                }
            }
        }
    }
#pragma unroll 1
    for (int p = 0; p < 2; p++) {
        double v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = p ? acc1[r] : acc0[r];
        uint32_t tl = tid;
        asm volatile("" : "+v"(tl));
        ntt_inverse_regs<AR, L>(v, s, ivt, m, tl);
        uint64_t *o = out + ((size_t)ct * 2 + p) * kn + (size_t)j * N;
#pragma unroll
        for (int r = 0; r < 16; r++) o[pass_index<L, SA, 0>(tl, r)] = AR::to_u64(v[r], m);
        __syncthreads();
    }
}

// Software-pipelined digit loop with two LDS images: the first pass (the only exchange that crosses waves) of digit g+1 is computed and
// written into image (g+1)&1 at the END of iteration g, the one workgroup barrier of an iteration sits at its START - a whole iteration
// of wave-local work (passes B, C, last pass, multiply-accumulate) after the writes it publishes, and every read of the image that is
// overwritten next lies a full iteration back.  One barrier per digit instead of two, and it no longer sits between a wave's own
// store and load of the same exchange.  Twiddles from L2 (two images + the table exceed the 160 KiB of LDS).
template <int F>
__global__ void __launch_bounds__(512, 1) k_ks_pipe(const uint64_t *__restrict__ target, const double *__restrict__ key, uint64_t *out, const double *tw_, double q, double qinv) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *s = reinterpret_cast<double *>(smem);
    const uint32_t k = 5, tid = threadIdx.x, ct = blockIdx.x / k, j = blockIdx.x % k;
    if constexpr ((F & F_PRIO) != 0) { if (tid < 256) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
    const AR::Mod m = {q, qinv};
    const size_t kn = (size_t)k * N;
    typedef ArPassA<AR> FW;
    typename FW::Tw fwt;
    const NTT_GLOBAL double *gtw = (const NTT_GLOBAL double *)tw_ + (size_t)j * 2 * N;
    fwt.w = gtw;
    ntt_load_pass_a<SA>(fwt, gtw);
    const AR::Tw ivt = {gtw + N};
    double acc0[16], acc1[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0; acc1[r] = 0; }
    uint64_t raw[16];
    auto load_raw = [&](uint32_t l) {
        const uint64_t *src = target + (size_t)ct * 3 * kn + 2 * kn + (size_t)l * N;
        uint32_t t0 = tid;
        asm volatile("" : "+v"(t0));
#pragma unroll
        for (int r = 0; r < 16; r++) raw[r] = src[pass_index<L, SA, 0>(t0, r)];
    };
    auto first_pass = [&](uint32_t g) {                  // digit g -> pass A -> image g & 1
        uint32_t tl = tid;
        asm volatile("" : "+v"(tl));
        double v[16];
        const int sh = 10 * (int)(g % 5);
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = AR::from_u64((raw[r] >> sh) & 1023);
        fwd_stages<FW, L, SA, 0>(v, fwt, m, tl);
        lds_put<double, L, SA, 0>(v, s + (g & 1) * ntt_lds_words(N), tl);
    };
    load_raw(0);
    first_pass(0);
    const double *kp = key;
    for (uint32_t g = 0; g < 25; g++, kp += 2 * kn) {
        __syncthreads();
        {
            uint32_t tl = tid;
            asm volatile("" : "+v"(tl));
            double *im = s + (g & 1) * ntt_lds_words(N);
            double v[16];
            lds_get<double, L, 4, SA>(v, im, tl);
            fwd_stages<FW, L, 4, SA>(v, fwt, m, tl);
            lds_put<double, L, 4, SA>(v, im, tl); ntt_wave_sync(); lds_get<double, L, 4, SA + 4>(v, im, tl);
            fwd_stages<FW, L, 4, SA + 4>(v, fwt, m, tl);
            lds_put<double, L, 4, SA + 4>(v, im, tl); ntt_wave_sync(); lds_get_tail<double, L>(v, im, tl);
            fwd_tail<FW, L>(v, fwt, m, tl);
            const double *k0 = kp + (size_t)j * N, *k1 = kp + kn + (size_t)j * N;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const uint32_t pos = tail_index<L>(tl, r);
                struct alignas(16) P2 { double a, b; };
                const P2 a = *reinterpret_cast<const P2 *>(k0 + pos), b = *reinterpret_cast<const P2 *>(k1 + pos);
                acc0[r] = __dadd_rn(acc0[r], AR::mulmod(v[r], a.a, m)); acc0[r + 1] = __dadd_rn(acc0[r + 1], AR::mulmod(v[r + 1], a.b, m));
                acc1[r] = __dadd_rn(acc1[r], AR::mulmod(v[r], b.a, m)); acc1[r + 1] = __dadd_rn(acc1[r + 1], AR::mulmod(v[r + 1], b.b, m));
            }
        }
        if (g + 1 < 25) {
            if ((g + 1) % 5 == 0) load_raw((g + 1) / 5);
            first_pass(g + 1);
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < 2; p++) {
        double v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = p ? acc1[r] : acc0[r];
        uint32_t tl = tid;
        asm volatile("" : "+v"(tl));
        ntt_inverse_regs<AR, L>(v, s, ivt, m, tl);
        uint64_t *o = out + ((size_t)ct * 2 + p) * kn + (size_t)j * N;
#pragma unroll
        for (int r = 0; r < 16; r++) o[pass_index<L, SA, 0>(tl, r)] = AR::to_u64(v[r], m);
        __syncthreads();
    }
}
template <int F> void run_pipe(const char *what, const uint64_t *tgt, const double *key, uint64_t *out, const double *tw, int cts) {
    const size_t lds = (size_t)ntt_lds_words(N) * 2 * 8;
    hipFuncSetAttribute((const void *)k_ks_pipe<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const double q = 8796092792833.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_ks_pipe<F>, dim3(cts * 5), dim3(512), lds, 0, tgt, key, out, tw, q, 1.0 / q);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_ks_pipe<F>, dim3(cts * 5), dim3(512), lds, 0, tgt, key, out, tw, q, 1.0 / q); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-64s : %7.3f ms\n", what, best);
}
template <int F> void run(const char *what, const uint64_t *tgt, const double *key, uint64_t *out, const double *tw, int cts) {
    const size_t lds = ((size_t)ntt_lds_words(N) * ((F & F_DB) ? 2 : 1) + ((F & (F_TWL | F_KLDS)) ? N : 0)) * 8;
    hipFuncSetAttribute((const void *)k_ks<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const double q = 8796092792833.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_ks<F>, dim3(cts * 5), dim3(512), lds, 0, tgt, key, out, tw, q, 1.0 / q);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_ks<F>, dim3(cts * 5), dim3(512), lds, 0, tgt, key, out, tw, q, 1.0 / q); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-64s : %7.3f ms\n", what, best);
}

int main() {
    const int cts = 845;
    const size_t kn = 5 * N;
    uint64_t *tgt, *out; double *key, *tw;
    hipMalloc(&tgt, (size_t)cts * 3 * kn * 8); hipMalloc(&out, (size_t)cts * 2 * kn * 8); hipMalloc(&key, (size_t)25 * 2 * kn * 8); hipMalloc(&tw, 5 * 2 * N * 8);
    std::vector<uint64_t> h((size_t)cts * 3 * kn);
    uint64_t st = 88172645463325252ull;
    for (auto &v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = st % 8796092792833ull; }
    hipMemcpy(tgt, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> kk((size_t)25 * 2 * kn);
    for (size_t i = 0; i < kk.size(); i++) kk[i] = (double)(h[i] % 8796092792833ull);
    hipMemcpy(key, kk.data(), kk.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(tw, kk.data(), 5 * 2 * N * 8, hipMemcpyHostToDevice);
    run<127>("full (keys, LDS, math, MAC, barriers, LDS twiddles, source loads)", tgt, key, out, tw, cts);
    run_pipe<0>("software-pipelined: first pass of digit g+1 before the barrier, 2 images", tgt, key, out, tw, cts);
    run_pipe<F_PRIO>("software-pipelined + waves 0-3 at priority 3", tgt, key, out, tw, cts);
    run<127 + F_PRIO>("full, waves 0-3 at priority 3, waves 4-7 at priority 0", tgt, key, out, tw, cts);
    run<127 - F_TWL>("full, twiddles from L2 instead of LDS", tgt, key, out, tw, cts);
    run<127 - F_TWL + F_DB>("two LDS images (one barrier per digit), twiddles from L2", tgt, key, out, tw, cts);
    run<127 - F_TWL + F_DB - F_KEYS>("two LDS images, twiddles from L2, no key loads", tgt, key, out, tw, cts);
    run<127 - F_TWL - F_BAR>("no barriers, twiddles from L2", tgt, key, out, tw, cts);
    run<127 - F_TWL + F_KLDS>("first key component prefetched into LDS, twiddles from L2", tgt, key, out, tw, cts);
    run<127 - F_KEYS>("no key loads", tgt, key, out, tw, cts);
    run<127 - F_BAR>("no barriers in the digit loop", tgt, key, out, tw, cts);
    run<127 - F_LDS>("no LDS exchanges", tgt, key, out, tw, cts);
    run<127 - F_LDS - F_BAR>("no LDS exchanges, no barriers", tgt, key, out, tw, cts);
    run<127 - F_LDS - F_BAR - F_KEYS>("no LDS exchanges, no barriers, no key loads", tgt, key, out, tw, cts);
    run<127 - F_MAC>("no MAC arithmetic (keys still loaded)", tgt, key, out, tw, cts);
    run<127 - F_MATH>("no transform arithmetic", tgt, key, out, tw, cts);
    run<F_MATH + F_MAC + F_TWL>("arithmetic only (LDS twiddles)", tgt, key, out, tw, cts);
    run<F_MATH + F_TWL>("transform arithmetic only", tgt, key, out, tw, cts);
    return 0;
}

// tools/store_flavour_probe.hip — the step kernels are WRITE-bound at 2^20 lanes (profiles/write_path_probe_r02.txt: the write half of
// CartPole's access pattern alone takes 5.0 us = 4.4 TB/s, the read half 2.8 us).  Does the store's cache-policy flavour move that?
// gfx950 global stores carry three bits — sc0, sc1, nt; the product uses `nt` (__builtin_nontemporal_store).  This probe writes CartPole's
// write pattern (4 state rows + reward as 16-byte stores, done as 4-byte stores) and the in-place copy (read + write) with every
// flavour, inline asm so that the bits are exactly the ones named.  HIP events over back-to-back launches, median of rounds.
//
//   bash tools/build_probes.sh store_flavour_probe && tools/build/store_flavour_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

struct Bufs { float *s[4]; int32_t *act; float *reward; uint8_t *done; };

// F: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0, 5 nt sc1, 6 nt sc0 sc1
template <int F> __device__ __forceinline__ void st16(float *p, f4 v) {
    if constexpr (F == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else if constexpr (F == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else if constexpr (F == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (F == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (F == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else if constexpr (F == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}
template <int F> __device__ __forceinline__ void st4(uint32_t *p, uint32_t v) {
    if constexpr (F == 0) asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else if constexpr (F == 1) asm volatile("global_store_dword %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else if constexpr (F == 2) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (F == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (F == 4) asm volatile("global_store_dword %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else if constexpr (F == 5) asm volatile("global_store_dword %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}

// READ: also read the four state rows and the actions first (non-temporal loads, the product's) — the in-place update
template <int F, bool READ> __global__ __launch_bounds__(256) void k(const Bufs b, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f4 v[4];
    float bump = 0.0f;
    if constexpr (READ) {
        for (int c = 0; c < 4; ++c) v[c] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(b.s[c] + i));
        const i4 a = __builtin_nontemporal_load(reinterpret_cast<const i4 *>(b.act + i));
        bump = (float)(a.x + a.y + a.z + a.w) * 1e-9f;
    } else {
        const float x = (float)(i & 1023);
        for (int c = 0; c < 4; ++c) v[c] = f4{x, x + 1.0f, x + 2.0f, x + 3.0f};
    }
    for (int c = 0; c < 4; ++c) st16<F>(b.s[c] + i, v[c] + bump);
    st16<F>(b.reward + i, f4{1.0f, 1.0f, 1.0f, 1.0f});
    st4<F>(reinterpret_cast<uint32_t *>(b.done + i), 0u);
}

template <int F, bool READ> static double run(const Bufs &b, int64_t n, hipStream_t s, int launches, int rounds) {
    const int grid = (int)((n / 4 + 255) / 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> us;
    for (int r = 0; r < rounds + 1; ++r) {
        CK(hipEventRecord(e0, s));
        for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k<F, READ>), dim3(grid), dim3(256), 0, s, b, n);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) us.push_back(ms * 1e3 / launches);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    std::sort(us.begin(), us.end());
    return us[us.size() / 2];
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const char *names[] = {"plain", "nt", "sc1", "sc0 sc1", "sc0", "sc1 nt", "sc0 sc1 nt"};
    for (int lg : {20, 21, 24}) {
        const int64_t n = (int64_t)1 << lg;
        // the product's layout: the four state rows in one allocation, stride n
        float *block;
        CK(hipMalloc(&block, (size_t)n * 4 * 4)); CK(hipMemset(block, 0, (size_t)n * 4 * 4));
        Bufs b{};
        for (int c = 0; c < 4; ++c) b.s[c] = block + c * n;
        CK(hipMalloc(&b.act, n * 4)); CK(hipMemset(b.act, 0, n * 4));
        CK(hipMalloc(&b.reward, n * 4)); CK(hipMalloc(&b.done, n));
        const int launches = lg >= 24 ? 50 : 400, rounds = 7;
        const double wB = 21.0 * n, rB = 20.0 * n;
        std::printf("lanes = 2^%d: write half %.1f MB, with the read half %.1f MB per launch; us per launch (GB/s)\n", lg, wB / 1e6, (wB + rB) / 1e6);
        double w[7], c[7];
        w[0] = run<0, false>(b, n, s, launches, rounds); c[0] = run<0, true>(b, n, s, launches, rounds);
        w[1] = run<1, false>(b, n, s, launches, rounds); c[1] = run<1, true>(b, n, s, launches, rounds);
        w[2] = run<2, false>(b, n, s, launches, rounds); c[2] = run<2, true>(b, n, s, launches, rounds);
        w[3] = run<3, false>(b, n, s, launches, rounds); c[3] = run<3, true>(b, n, s, launches, rounds);
        w[4] = run<4, false>(b, n, s, launches, rounds); c[4] = run<4, true>(b, n, s, launches, rounds);
        w[5] = run<5, false>(b, n, s, launches, rounds); c[5] = run<5, true>(b, n, s, launches, rounds);
        w[6] = run<6, false>(b, n, s, launches, rounds); c[6] = run<6, true>(b, n, s, launches, rounds);
        for (int f = 0; f < 7; ++f)
            std::printf("   %-11s write only %7.3f us (%6.0f)    read + write in place %7.3f us (%6.0f)\n", names[f], w[f], wB / (w[f] * 1e-6) / 1e9, c[f],
                        (wB + rB) / (c[f] * 1e-6) / 1e9);
        CK(hipFree(block)); CK(hipFree(b.act)); CK(hipFree(b.reward)); CK(hipFree(b.done));
    }
    return 0;
}

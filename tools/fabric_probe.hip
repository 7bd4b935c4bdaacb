// fabric_probe: which half of the step kernel's traffic sets its floor?
//
// The CartPole step at 2^20 lanes reads 20 B and writes 21 B per lane, 16 bytes per thread per stream (four lanes per
// thread), and a copy-only kernel with that pattern takes as long as the real kernel.  This probe splits the pattern:
// the READ half alone (4 state streams + actions, folded into a value that is stored only if it is NaN), the WRITE half
// alone (4 state streams + reward + done, values from the lane index), and both.  Same launch shape, same 16-byte
// accesses, temporal and non-temporal; HIP events over back-to-back launches, median of rounds.
//
//   hipcc -O3 --offload-arch=gfx950 tools/fabric_probe.hip -o tools/fabric_probe && tools/fabric_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

struct Bufs { float *s[4]; int32_t *act; float *reward; uint8_t *done; float *sink; };

template <bool NT> __device__ inline f4 ld(const float *p) { const f4 *q = reinterpret_cast<const f4 *>(p); return NT ? __builtin_nontemporal_load(q) : *q; }
template <bool NT> __device__ inline i4 ldi(const int32_t *p) { const i4 *q = reinterpret_cast<const i4 *>(p); return NT ? __builtin_nontemporal_load(q) : *q; }
template <bool NT> __device__ inline void st(float *p, f4 v) { f4 *q = reinterpret_cast<f4 *>(p); if (NT) __builtin_nontemporal_store(v, q); else *q = v; }
template <bool NT> __device__ inline void stb(uint8_t *p, uint32_t v) { uint32_t *q = reinterpret_cast<uint32_t *>(p); if (NT) __builtin_nontemporal_store(v, q); else *q = v; }

// MODE 1 = read half, 2 = write half, 3 = both (copy)
template <int MODE, bool NT> __global__ __launch_bounds__(256) void k(const Bufs b, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f4 v[4];
    i4 a = {0, 0, 0, 0};
    if (MODE & 1) {
        for (int c = 0; c < 4; ++c) v[c] = ld<NT>(b.s[c] + i);
        a = ldi<NT>(b.act + i);
    } else {
        const float x = (float)(i & 1023);
        for (int c = 0; c < 4; ++c) v[c] = f4{x, x + 1.0f, x + 2.0f, x + 3.0f};
    }
    if (MODE & 2) {
        const f4 bump = {(float)a.x, (float)a.y, (float)a.z, (float)a.w};
        for (int c = 0; c < 4; ++c) st<NT>(b.s[c] + i, v[c] + bump * 0.0f);
        st<NT>(b.reward + i, f4{1.0f, 1.0f, 1.0f, 1.0f});
        stb<NT>(b.done + i, (uint32_t)(a.x & 1) * 0x01010101u);
    } else {
        f4 acc = v[0] + v[1] + v[2] + v[3];
        float r = acc.x + acc.y + acc.z + acc.w + (float)(a.x + a.y + a.z + a.w);
        if (r != r) b.sink[0] = r;           // never true for finite data: keeps the loads alive without a store
    }
}

template <int MODE, bool NT> static double run(const Bufs &b, int64_t n, hipStream_t s, int launches, int rounds) {
    const int grid = (int)((n / 4 + 255) / 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> us;
    for (int r = 0; r < rounds + 1; ++r) {
        CK(hipEventRecord(e0, s));
        for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(256), 0, s, b, n);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) us.push_back(ms * 1e3 / launches);
    }
    std::sort(us.begin(), us.end());
    return us[us.size() / 2];
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    std::printf("## tools/fabric_probe: read half / write half / both of the CartPole step's access pattern (us per launch, median)\n");
    for (int lg : {18, 20, 22, 24, 27}) {
        const int64_t n = (int64_t)1 << lg;
        Bufs b{};
        for (int c = 0; c < 4; ++c) { CK(hipMalloc(&b.s[c], n * 4)); CK(hipMemset(b.s[c], 0, n * 4)); }
        CK(hipMalloc(&b.act, n * 4)); CK(hipMemset(b.act, 0, n * 4));
        CK(hipMalloc(&b.reward, n * 4)); CK(hipMalloc(&b.done, n)); CK(hipMalloc(&b.sink, 64));
        const int launches = lg >= 24 ? 20 : 200, rounds = 9;
        const double rB = 20.0 * n, wB = 21.0 * n;
        std::printf("lanes=2^%d  read half %.1f MB, write half %.1f MB per launch\n", lg, rB / 1e6, wB / 1e6);
        struct Row { const char *name; double us; double bytes; };
        Row rows[] = {
            {"read  half, temporal    ", run<1, false>(b, n, s, launches, rounds), rB},
            {"read  half, non-temporal", run<1, true>(b, n, s, launches, rounds), rB},
            {"write half, temporal    ", run<2, false>(b, n, s, launches, rounds), wB},
            {"write half, non-temporal", run<2, true>(b, n, s, launches, rounds), wB},
            {"both,       temporal    ", run<3, false>(b, n, s, launches, rounds), rB + wB},
            {"both,       non-temporal", run<3, true>(b, n, s, launches, rounds), rB + wB},
        };
        for (const Row &r : rows) std::printf("  %s  %8.3f us   %7.1f GB/s\n", r.name, r.us, r.bytes / (r.us * 1e-6) / 1e9);
        // the same copy with the four state streams inside ONE allocation, `stride` floats apart (the product's layout)
        for (int64_t pad : {(int64_t)0, (int64_t)64, (int64_t)1024, (int64_t)(16384 + 64), (int64_t)(65536 + 1024 + 64)}) {
            if (lg > 24) break;
            const int64_t stride = n + pad;
            float *block;
            CK(hipMalloc(&block, stride * 4 * 4)); CK(hipMemset(block, 0, stride * 4 * 4));
            Bufs c = b;
            for (int q = 0; q < 4; ++q) c.s[q] = block + q * stride;
            const double us = run<3, true>(c, n, s, launches, rounds);
            const double usw = run<2, true>(c, n, s, launches, rounds);
            std::printf("  both / write half, nt, one block, stride = n + %-6lld  %8.3f us %7.1f GB/s   /  %8.3f us\n", (long long)pad, us, (rB + wB) / (us * 1e-6) / 1e9, usw);
            CK(hipFree(block));
        }
        // every stream (4 state, action, reward, done) inside ONE allocation, stream k at k * (4 n + pad) bytes: how does
        // the copy depend on the address relation between the streams?
        if (lg == 20 || lg == 22) {
            std::printf("  separate allocations: s0 %p s1 %p s2 %p s3 %p act %p reward %p done %p\n", (void *)b.s[0], (void *)b.s[1],
                        (void *)b.s[2], (void *)b.s[3], (void *)b.act, (void *)b.reward, (void *)b.done);
            const int64_t pads[] = {0, 256, 512, 1024, 2048, 4096, 4096 + 256, 8192, 16384, 32768, 65536, 65536 + 4096, 131072, 262144,
                                    262144 + 4096, 524288, 1048576, 1048576 + 4096, 2097152 + 4096};
            for (int64_t pad : pads) {
                const int64_t pitch = n * 4 + pad;
                char *block;
                CK(hipMalloc(&block, pitch * 7)); CK(hipMemset(block, 0, pitch * 7));
                Bufs c = b;
                for (int q = 0; q < 4; ++q) c.s[q] = reinterpret_cast<float *>(block + q * pitch);
                c.act = reinterpret_cast<int32_t *>(block + 4 * pitch);
                c.reward = reinterpret_cast<float *>(block + 5 * pitch);
                c.done = reinterpret_cast<uint8_t *>(block + 6 * pitch);
                const double us = run<3, true>(c, n, s, launches, rounds);
                const double ust = run<3, false>(c, n, s, launches, rounds);
                std::printf("  all streams in one block, pitch = 4n + %-8lld  nt %8.3f us %7.1f GB/s   temporal %8.3f us\n", (long long)pad, us,
                            (rB + wB) / (us * 1e-6) / 1e9, ust);
                CK(hipFree(block));
            }
        }
        for (int c = 0; c < 4; ++c) CK(hipFree(b.s[c]));
        CK(hipFree(b.act)); CK(hipFree(b.reward)); CK(hipFree(b.done)); CK(hipFree(b.sink));
    }
    return 0;
}

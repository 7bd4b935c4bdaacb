// issue_rate_probe.hip — round 6: what a counted VALU instruction costs its SIMD on gfx950, for the classes the fused rollouts are
// made of, so that the rollout's VALU-issue floor (bench.py valu_roofline) is priced on measured rates instead of "4 clocks each":
//   * v_fma_f32 with a FULL exec mask, and with only the first A lanes active (A = 48, 32, 16, 11, 1): does the SIMD skip the
//     16-lane passes whose lanes are all inactive?  (the wave-compacted reset's Philox pass runs with ~11 of 64 lanes)
//   * v_mad_u64_u32 (the Philox round's product), v_xor3_b32, v_cndmask_b32, v_fma_f64, v_rcp_f32
// 8 waves per SIMD, 16 independent chains per thread, cycles per wave-instruction at the 2.4 GHz the device reports.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize tools/issue_rate_probe.hip -o tools/build/issue_rate_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int ACC = 16;

// which lanes run the loop: MODE 0 the first ACTIVE lanes; 1 every fourth lane (16 lanes, four in each 16-lane quarter); 2 lanes 16 .. 31 only;
// 3 the first ACTIVE lanes AND lane 63
template <int ACTIVE, int MODE = 0>
__global__ __launch_bounds__(256) void k_fma(float *out, int iters, float a, float b) {
    float x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) x[i] = (float)(threadIdx.x + i);
    const int lane = threadIdx.x & 63;
    const bool on = MODE == 0 ? lane < ACTIVE : MODE == 1 ? (lane & 3) == 0 : MODE == 2 ? (lane >= 16 && lane < 32) : (lane < ACTIVE || lane == 63);
    if (on) {                                          // the loop runs under this exec mask
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < ACC; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void k_mad64(float *out, int iters, float a, float b) {
    uint32_t x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) x[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) { const uint64_t p = (uint64_t)0xD2511F53u * x[i]; x[i] = (uint32_t)(p >> 32) ^ (uint32_t)p; }   // 1 mad + 1 xor
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i];
    if (s == 12345u) out[0] = (float)s;
}

__global__ __launch_bounds__(256) void k_xor(float *out, int iters, float a, float b) {
    uint32_t x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) x[i] = threadIdx.x * 2654435761u + i;
    const uint32_t k = (uint32_t)iters * 77u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) { x[i] = (x[i] ^ k) + (uint32_t)it; }      // 2 integer ops (or one fused)
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i];
    if (s == 12345u) out[0] = (float)s;
}

__global__ __launch_bounds__(256) void k_fma64(float *out, int iters, float a, float b) {
    double x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) x[i] = (double)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) x[i] = __builtin_fma(x[i], (double)a, (double)b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i];
    if (s == 12345.678) out[0] = (float)s;
}

__global__ __launch_bounds__(256) void k_rcp(float *out, int iters, float a, float b) {
    float x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) x[i] = (float)(threadIdx.x + i) + 1.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) x[i] = __builtin_amdgcn_rcpf(x[i]);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;
}

// Do instructions of DIFFERENT classes overlap on a SIMD?  Each thread runs 8 float32 FMA chains and 8 chains of a second class per loop trip
// (independent of each other); if the classes share one issue slot the time is the sum of the two halves run alone, if the second class has a
// pipe of its own it is closer to the longer half.  MODE 0: both; 1: only the float32 half; 2: only the second half.
// CLASS 0: v_fma_f64, 1: v_mad_u64_u32, 2: v_rcp_f32, 3: v_cvt_f64_f32 + v_cmp_gt_f64-ish (conversion + compare, the `done` arithmetic)
template <int CLASS, int MODE>
__global__ __launch_bounds__(256) void k_mix(float *out, int iters, float a, float b) {
    float x[8];
    double y[8];
    uint32_t z[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = (float)(threadIdx.x + i); y[i] = (double)(threadIdx.x + i) + 0.5; z[i] = threadIdx.x * 2654435761u + i; }
    const double ad = (double)a, bd = (double)b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE != 2) x[i] = __builtin_fmaf(x[i], a, b);
            if (MODE != 1) {
                if (CLASS == 0) y[i] = __builtin_fma(y[i], ad, bd);
                else if (CLASS == 1) { const uint64_t p = (uint64_t)0xD2511F53u * z[i]; z[i] = (uint32_t)(p >> 32) + (uint32_t)p; }     // mad (the add folds into it)
                else if (CLASS == 2) { float r = __uint_as_float(z[i] | 0x3f800000u); r = __builtin_amdgcn_rcpf(r); z[i] = __float_as_uint(r); }
                else { const double d = (double)__uint_as_float(z[i] | 0x3f000000u); z[i] += d > y[i] ? 3u : 1u; }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + (float)y[i] + (float)z[i];
    if (s == 12345.678f) out[0] = s;
}

// the engine clock under load: s_memtime (clock64: shader clock cycles) against s_memrealtime (wall_clock64: the constant 100 MHz counter)
// around a long full-mask FMA loop, read by lane 0 of every 1024th workgroup while the whole chip runs the same loop
__global__ __launch_bounds__(256) void k_clock(float *out, int iters, float a, float b, unsigned long long *ticks) {
    float x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) x[i] = (float)(threadIdx.x + i);
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && (blockIdx.x & 1023) == 0) { ticks[2 * (blockIdx.x >> 10)] = c1 - c0; ticks[2 * (blockIdx.x >> 10) + 1] = w1 - w0; }
}

template <class K>
static double run(K kern, int blocks, int iters, float *d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, 1e-9f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001f, 1e-9f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    printf("device %s, %d CUs, clockRate %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
    float *d;
    (void)hipMalloc(&d, 64);
    const int iters = 4096, wps = 8;
    const int blocks = cus * wps;            // 4 waves per block = one per SIMD: wps blocks per CU = wps waves per SIMD
    auto report = [&](const char *name, double ms, double instr_per_iter_chain) {
        const double wave_instr_per_simd = (double)wps * iters * ACC * instr_per_iter_chain;
        printf("%-34s %8.3f ms  %6.2f SIMD-cycles per wave-instruction\n", name, ms, ms * 1e-3 * ghz * 1e9 / wave_instr_per_simd);
    };
    {
        unsigned long long *t;
        (void)hipMalloc(&t, 64 * sizeof(unsigned long long));
        (void)hipMemset(t, 0, 64 * sizeof(unsigned long long));
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, d, iters * 4, 1.0000001f, 1e-9f, t);
            (void)hipDeviceSynchronize();
            unsigned long long h[4];
            (void)hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
            int wall_khz = 100000;
            (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
            printf("engine clock under load: %llu shader cycles in %llu wall ticks at %d kHz = %.3f GHz (second sample %.3f GHz)\n", h[0], h[1], wall_khz,
                   (double)h[0] / ((double)h[1] / (wall_khz * 1e3)) / 1e9, (double)h[2] / ((double)h[3] / (wall_khz * 1e3)) / 1e9);
        }
    }
    report("v_fma_f32   64 of 64 lanes", run(k_fma<64>, blocks, iters, d), 1);
    report("v_fma_f32   48 of 64 lanes", run(k_fma<48>, blocks, iters, d), 1);
    report("v_fma_f32   32 of 64 lanes", run(k_fma<32>, blocks, iters, d), 1);
    report("v_fma_f32   16 of 64 lanes", run(k_fma<16>, blocks, iters, d), 1);
    report("v_fma_f32   11 of 64 lanes", run(k_fma<11>, blocks, iters, d), 1);
    report("v_fma_f32    1 of 64 lanes", run(k_fma<1>, blocks, iters, d), 1);
    report("v_fma_f32   17 of 64 lanes", run(k_fma<17>, blocks, iters, d), 1);
    report("v_fma_f32   24 of 64 lanes", run(k_fma<24>, blocks, iters, d), 1);
    report("v_fma_f32   every 4th lane (16)", run(k_fma<16, 1>, blocks, iters, d), 1);
    report("v_fma_f32   lanes 16..31 only", run(k_fma<16, 2>, blocks, iters, d), 1);
    report("v_fma_f32   lanes 0..10 and 63", run(k_fma<11, 3>, blocks, iters, d), 1);
    {   // the same at ONE wave per SIMD and at 2: is it a matter of how many waves share the SIMD?
        const int b1 = cus * 1, b2 = cus * 2;
        auto rep2 = [&](const char *name, double ms, int w) {
            printf("%-34s %8.3f ms  %6.2f SIMD-cycles per wave-instruction (%d wave(s) per SIMD)\n", name, ms, ms * 1e-3 * ghz * 1e9 / ((double)w * iters * ACC), w);
        };
        rep2("v_fma_f32   64 of 64 lanes", run(k_fma<64>, b1, iters, d), 1);
        rep2("v_fma_f32   11 of 64 lanes", run(k_fma<11>, b1, iters, d), 1);
        rep2("v_fma_f32   64 of 64 lanes", run(k_fma<64>, b2, iters, d), 2);
        rep2("v_fma_f32   11 of 64 lanes", run(k_fma<11>, b2, iters, d), 2);
    }
    {
        const char *cls[4] = {"v_fma_f64", "v_mad_u64_u32", "v_rcp_f32 (+ or / bitcast)", "cvt_f64_f32 + cmp_f64 + cndmask + add"};
        auto three = [&](int c, double both, double f32only, double other) {
            printf("mix 8 x v_fma_f32 + 8 x %-38s both %7.3f ms   float32 half alone %7.3f   other half alone %7.3f   sum of halves %7.3f   overlap %4.0f %% of the shorter half\n",
                   cls[c], both, f32only, other, f32only + other, 100.0 * (f32only + other - both) / (f32only < other ? f32only : other));
        };
        three(0, run(k_mix<0, 0>, blocks, iters, d), run(k_mix<0, 1>, blocks, iters, d), run(k_mix<0, 2>, blocks, iters, d));
        three(1, run(k_mix<1, 0>, blocks, iters, d), run(k_mix<1, 1>, blocks, iters, d), run(k_mix<1, 2>, blocks, iters, d));
        three(2, run(k_mix<2, 0>, blocks, iters, d), run(k_mix<2, 1>, blocks, iters, d), run(k_mix<2, 2>, blocks, iters, d));
        three(3, run(k_mix<3, 0>, blocks, iters, d), run(k_mix<3, 1>, blocks, iters, d), run(k_mix<3, 2>, blocks, iters, d));
    }
    report("v_mad_u64_u32 + v_xor (per pair)", run(k_mad64, blocks, iters, d), 1);
    report("xor + add (per pair)", run(k_xor, blocks, iters, d), 1);
    report("v_fma_f64", run(k_fma64, blocks, iters, d), 1);
    report("v_rcp_f32", run(k_rcp, blocks, iters, d), 1);
    return 0;
}

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

template <int ACTIVE>
__global__ __launch_bounds__(256) void k_fma(float *out, int iters, float a, float b) {
    float x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) x[i] = (float)(threadIdx.x + i);
    if ((threadIdx.x & 63) < ACTIVE) {                 // the loop runs under this exec mask
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
    report("v_fma_f32   64 of 64 lanes", run(k_fma<64>, blocks, iters, d), 1);
    report("v_fma_f32   48 of 64 lanes", run(k_fma<48>, blocks, iters, d), 1);
    report("v_fma_f32   32 of 64 lanes", run(k_fma<32>, blocks, iters, d), 1);
    report("v_fma_f32   16 of 64 lanes", run(k_fma<16>, blocks, iters, d), 1);
    report("v_fma_f32   11 of 64 lanes", run(k_fma<11>, blocks, iters, d), 1);
    report("v_fma_f32    1 of 64 lanes", run(k_fma<1>, blocks, iters, d), 1);
    report("v_mad_u64_u32 + v_xor (per pair)", run(k_mad64, blocks, iters, d), 1);
    report("xor + add (per pair)", run(k_xor, blocks, iters, d), 1);
    report("v_fma_f64", run(k_fma64, blocks, iters, d), 1);
    report("v_rcp_f32", run(k_rcp, blocks, iters, d), 1);
    return 0;
}

// valu_probe.hip — ground truth for the Acrobot kernel's ALU budget on gfx950: how many cycles does a wave64 FP32 VALU
// instruction occupy its SIMD, and do the packed forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) move two floats per
// lane in the same slot?   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o tools/valu_probe tools/valu_probe.hip   (SLP off so that the scalar rows stay scalar)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int ACC>
__global__ __launch_bounds__(256) void k_fma(float *out, int iters, float a, float b) {
    float x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) x[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;
}

template <int ACC>
__global__ __launch_bounds__(256) void k_pkfma(float *out, int iters, float a, float b) {
    f2 x[ACC];
    const f2 av = {a, a * 1.0001f}, bv = {b, b * 0.9999f};
#pragma unroll
    for (int i = 0; i < ACC; ++i) x[i] = f2{(float)(threadIdx.x + i), (float)(threadIdx.x - i)};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
    }
    f2 s = {0, 0};
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i];
    if (s.x + s.y == 12345.678f) out[0] = s.x;
}

template <int ACC>
__global__ __launch_bounds__(256) void k_muladd(float *out, int iters, float a, float b) {   // separate mul + add (contract off)
    float x[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) x[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) { float t = x[i] * a; asm volatile("" : "+v"(t)); x[i] = t + b; }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;
}

template <int ACC>
__global__ __launch_bounds__(256) void k_pkmuladd(float *out, int iters, float a, float b) {
    f2 x[ACC];
    const f2 av = {a, a * 1.0001f}, bv = {b, b * 0.9999f};
#pragma unroll
    for (int i = 0; i < ACC; ++i) x[i] = f2{(float)(threadIdx.x + i), (float)(threadIdx.x - i)};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) { f2 t = x[i] * av; asm volatile("" : "+v"(t)); x[i] = t + bv; }
    }
    f2 s = {0, 0};
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += x[i];
    if (s.x + s.y == 12345.678f) out[0] = s.x;
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
    return ms * 1e-3;
}

int main() {
    float *d;
    (void)hipMalloc(&d, 1024);
    const int iters = 20000;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const double clk = p.clockRate * 1e3;
    printf("device %s, %d CUs, clockRate %.0f MHz\n", p.name, p.multiProcessorCount, clk / 1e6);
    for (int wps : {1, 2, 4, 8}) {                         // waves per SIMD
        const int blocks = 256 * wps;                      // 256 CUs x (4 waves per block) -> wps waves on each of the 4 SIMDs
        const double waves_per_simd = wps;
        struct { const char *name; double t; double instr_per_wave; double flop_per_lane; } rows[] = {
            {"v_fma_f32      x16 indep", run(k_fma<16>, blocks, iters, d), 16.0 * iters, 2.0 * 16 * iters},
            {"v_pk_fma_f32   x8 indep ", run(k_pkfma<8>, blocks, iters, d), 8.0 * iters, 2.0 * 16 * iters},
            {"v_pk_fma_f32   x16 indep", run(k_pkfma<16>, blocks, iters, d), 16.0 * iters, 2.0 * 32 * iters},
            {"v_mul+v_add    x16 indep", run(k_muladd<16>, blocks, iters, d), 32.0 * iters, 2.0 * 16 * iters},
            {"v_pk_mul+pk_add x8 indep", run(k_pkmuladd<8>, blocks, iters, d), 16.0 * iters, 2.0 * 16 * iters},
            {"v_fma_f32      x2 indep ", run(k_fma<2>, blocks, iters, d), 2.0 * iters, 2.0 * 2 * iters},
            {"v_pk_fma_f32   x1 chain ", run(k_pkfma<1>, blocks, iters, d), 1.0 * iters, 2.0 * 2 * iters},
        };
        for (auto &r : rows) {
            const double cyc_per_instr = r.t * clk / (r.instr_per_wave * waves_per_simd);
            const double tflops = r.flop_per_lane * 64.0 * (double)blocks * 4.0 / r.t / 1e12;
            printf("waves/SIMD %d  %s  %.3f ms  %.2f SIMD-cycles per wave-instruction  %.1f TFLOP/s\n", wps, r.name, r.t * 1e3, cyc_per_instr, tflops);
        }
    }
    return 0;
}

// acrobot_alu_probe.hip — the Acrobot step's arithmetic alone (state in registers, no memory traffic): one env per lane on
// scalar FP32 vs two envs per lane on packed FP32 (envs.hpp step_observe / step_observe_x2), at several occupancies.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I gym.net_amd/csrc -o tools/acrobot_alu_probe tools/acrobot_alu_probe.hip
#include "envs.hpp"
#include <cstdio>
using namespace gymnet;

__global__ __launch_bounds__(256) void k1(float *out, int iters) {
    float s[4] = {0.01f * threadIdx.x, -0.02f * threadIdx.x, 0.3f, -0.2f}, o[6];
    float acc = 0;
    for (int t = 0; t < iters; ++t) {
        float r; bool d;
        Acrobot::step_observe(s, (t + threadIdx.x) % 3, r, d, o);
        acc += r + o[0];
    }
    if (acc == 12345.0f) out[0] = acc + s[0];
}
__global__ __launch_bounds__(256) void k2(float *out, int iters) {
    float s[4][2] = {{0.01f * threadIdx.x, 0.011f * threadIdx.x}, {-0.02f * threadIdx.x, 0.5f}, {0.3f, 0.1f}, {-0.2f, 0.4f}}, o[6][2];
    float acc = 0;
    for (int t = 0; t < iters; ++t) {
        float r[2]; bool d[2]; int a[2] = {(int)((t + threadIdx.x) % 3), (int)((t + 2 * threadIdx.x) % 3)};
        Acrobot::step_observe_x2(s, a, r, d, o);
        acc += r[0] + r[1] + o[0][0] + o[0][1];
    }
    if (acc == 12345.0f) out[0] = acc + s[0][0];
}
template <class K> static double run(K kern, int blocks, int iters, float *d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e-3;
}
int main() {
    float *d; (void)hipMalloc(&d, 1024);
    const int iters = 400;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        const double t1 = run(k1, blocks, iters, d), t2 = run(k2, blocks, iters, d);
        const double lanes = (double)blocks * 256;
        // time to advance 2^20 envs by one step at this occupancy if the arithmetic were all there is
        printf("waves/SIMD %d  scalar %.3f ns/env-step/SIMD-slot -> %.2f us per 2^20 env-steps   packed x2 -> %.2f us per 2^20 env-steps\n", wps,
               t1 / iters * 1e9, t1 / iters / lanes * (1 << 20) * 1e6, t2 / iters / (2 * lanes) * (1 << 20) * 1e6);
    }
    return 0;
}

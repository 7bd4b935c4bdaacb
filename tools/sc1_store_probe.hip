// tools/sc1_store_probe.hip — the shipped CartPole step kernel with its non-temporal 16-byte stores replaced by write-through (`sc1`) stores
// (lanes.hpp, GYMNET_PROBE_STORE_SC1: this translation unit only), against the same kernel as shipped — built twice from this file:
//   hipcc ... tools/sc1_store_probe.hip -o tools/build/sc1_store_probe_nt
//   hipcc ... -DGYMNET_PROBE_STORE_SC1 tools/sc1_store_probe.hip -o tools/build/sc1_store_probe_sc1
// (tools/store_flavour_probe.hip: a pure in-place copy of 2^21 lanes runs 8 % faster with sc1 than with nt stores, and slower at 2^20.)
//   usage: sc1_store_probe_* [launches = 1000] [rounds = 5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#include "../gym.net_amd/csrc/step_kernels.hpp"
#include "../gym.net_amd/csrc/envs.hpp"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(2); } } while (0)
using namespace gymnet;

int main(int argc, char **argv) {
    const int launches = argc > 1 ? std::atoi(argv[1]) : 1000, rounds = argc > 2 ? std::atoi(argv[2]) : 5;
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
#ifdef GYMNET_PROBE_STORE_SC1
    const char *what = "sc1 (write-through) 16-byte stores";
#else
    const char *what = "nt 16-byte stores (as shipped)";
#endif
    constexpr int64_t kRing = 16;
    for (int lg : {20, 21, 22, 23}) {
        const int64_t n = (int64_t)1 << lg;
        float *state; int32_t *action; float *reward; uint8_t *done; uint64_t *tick2;
        HIP_OK(hipMalloc(&state, (size_t)4 * n * 4)); HIP_OK(hipMalloc(&action, (size_t)kRing * n * 4)); HIP_OK(hipMalloc(&reward, (size_t)n * 4));
        HIP_OK(hipMalloc(&done, (size_t)n)); HIP_OK(hipMalloc(&tick2, 16));
        HIP_OK(hipMemsetAsync(state, 0, (size_t)4 * n * 4, st)); HIP_OK(hipMemsetAsync(tick2, 0, 16, st));
        std::vector<uint32_t> act((size_t)kRing * n);
        for (int64_t i = 0; i < kRing * n; ++i) { uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x5EED; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; act[i] = (uint32_t)((z >> 40) & 1u); }
        HIP_OK(hipMemcpyAsync(action, act.data(), act.size() * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipStreamSynchronize(st));
        StepArgsT<float> a{};
        a.state = state; a.state_out = state; a.obs = state; a.obs_in = state; a.reward = reward; a.done = done; a.tick2 = tick2;
        a.n = n; a.state_stride = n; a.obs_stride = n; a.seed = 0x5EED;
        uint64_t tick = 0;
        const int L = lg >= 23 ? launches / 4 : launches;
        for (int mask : {15, 12}) {
            const LaunchCfg cfg{4, 256, mask, 0, 1, 1, 0};
            std::vector<double> us;
            for (int r = 0; r < rounds + 1; ++r) {
                hipEvent_t e0, e1;
                HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
                HIP_OK(hipEventRecord(e0, st));
                for (int i = 0; i < L; ++i) {
                    a.parity = (int32_t)(tick & 1); a.cparity = a.parity; a.action = action + (int64_t)(tick % kRing) * n;
                    HIP_OK((launch_step_env<CartPole>(true, false, a, cfg, st)));
                    ++tick;
                }
                HIP_OK(hipEventRecord(e1, st)); HIP_OK(hipEventSynchronize(e1));
                float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
                HIP_OK(hipEventDestroy(e0)); HIP_OK(hipEventDestroy(e1));
                if (r) us.push_back((double)ms * 1000.0 / L);
            }
            std::sort(us.begin(), us.end());
            std::printf("%-36s 2^%d lanes, mask %2d: %8.3f us per launch = %6.3f per 2^20 lanes  [%.3f, %.3f]\n", what, lg, mask, us[us.size() / 2],
                        us[us.size() / 2] * 1048576.0 / n, us.front(), us.back());
        }
        HIP_OK(hipFree(state)); HIP_OK(hipFree(action)); HIP_OK(hipFree(reward)); HIP_OK(hipFree(done)); HIP_OK(hipFree(tick2));
    }
    return 0;
}

// env_cartpole.hip — the step / fused-rollout / reset kernels of step_kernels.hpp instantiated for CartPole:
// CartPole-v1 in float32 (CartPoleEnv.cs:24-67,137-186), the structure-of-arrays hot path.  One translation unit per env so the build compiles them side by side.
#include "step_kernels.hpp"

#include "envs.hpp"

GYMNET_DEFINE_ENV(cartpole, gymnet::CartPole)

// env_cartpole64.hip — the step / fused-rollout / reset kernels of step_kernels.hpp instantiated for CartPole64:
// GYMNET_FLAG_F64 — CartPole in the reference's own binary64 arithmetic (CartPoleEnv.cs:141-166,185).  One translation unit per env so the build compiles them side by side.
#include "step_kernels.hpp"

#include "cartpole64.hpp"

GYMNET_DEFINE_ENV(cartpole64, gymnet::CartPole64)

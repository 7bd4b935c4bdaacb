// env_mountaincar.hip — the step / fused-rollout / reset kernels of step_kernels.hpp instantiated for MountainCar:
// MountainCar-v0 (absent from the reference; upstream gym).  One translation unit per env so the build compiles them side by side.
#include "step_kernels.hpp"

#include "envs.hpp"

GYMNET_DEFINE_ENV(mountaincar, gymnet::MountainCar)

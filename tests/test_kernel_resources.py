"""The register budget the launch policy relies on, asserted from the gfx950 assembly (no GPU): how many waves per SIMD each bench kernel can
hold decides whether a 2^20-lane launch is ONE generation of waves (docs/ledger.md §4d, profiles/occupancy_hints_r05.txt), and a spilling step
kernel would be a silent slowdown.  CartPole in both state scalars (the headline and the reference-arithmetic kernel); ~1 minute of hipcc."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.timeout(900)
def test_bench_kernels_hold_the_waves_per_simd_the_policy_assumes():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    import kernel_resources
    k = kernel_resources.collect(("cartpole", "cartpole64"))
    head = k["step_kernel<CartPole,4,true,false,15,1>"]                      # the headline: 4096 waves at 2^20 lanes = 4 per SIMD
    assert head["occupancy"] == 8 and head["scratch"] == 0
    for name, v in k.items():
        if name.startswith("step_kernel<") or name.startswith("step_kernel_pipe"):
            assert v["scratch"] == 0, (name, v)                              # no step kernel spills
    # float32 rollouts at 2^20 lanes are 4096 four-lane waves: every variant must hold FOUR per SIMD (the bookkeeping ones need the hint)
    for name, v in k.items():
        if name.startswith("rollout_kernel<CartPole,4,"):
            assert v["occupancy"] >= 4 and v["scratch"] <= 160, (name, v)
    # the float64 four-pair kernel: two waves per SIMD, no spills (the policy's window closes at exactly two waves per SIMD)
    f64 = k["step_kernel_pipe2<CartPole64,4,true,15>"]
    assert f64["occupancy"] == 2 and f64["scratch"] == 0
    assert k["step_kernel_pipe2<CartPole64,2,true,15>"]["occupancy"] >= 3    # 3072 waves at 3 * 2^18 lanes in one generation
    assert k["step_kernel<CartPole64,2,true,false,15,1>"]["occupancy"] >= 4

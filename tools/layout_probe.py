"""layout_probe: does the address relation between the state streams matter for the SHIPPED step kernel?
CartPole, 2^20 lanes, state placed in a caller-provided block (d_ext_obs) whose stream stride is n + pad floats;
reward / done stay the library's own allocations.  us per step by HIP events over one 2048-step rollout, median of 5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
n = int(os.environ.get("N", 1 << 20))
ring, K = 64, 2048
env_name = os.environ.get("ENV", "CartPole-v1")
actions = torch.empty((ring, n), dtype=torch.int32, device=dev)


def run(env):
    for t in range(ring):
        env.SampleActionsDevice(actions[t].data_ptr(), seed=1, tick=t)
    env.ResetDevice()
    env.RolloutDevice(actions.data_ptr(), 256, n, ring)
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        env.RolloutDevice(actions.data_ptr(), K, n, ring)
        e1.record(stream)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / K)
    return sorted(out)[2]


with pkg.VectorEnv(env_name, n, device=0, seed=3, auto_reset=True, stream=stream.cuda_stream) as e:
    print(f"{env_name} n={n}: library-owned state           {run(e):7.3f} us/step")
for pad in (0, 64, 128, 256, 512, 1024, 2048, 4096, 16384, 16448, 65536 + 1024):
    stride = n + pad
    buf = torch.zeros(4 * stride + 1024, dtype=torch.float32, device=dev)
    with pkg.VectorEnv(env_name, n, device=0, seed=3, auto_reset=True, stream=stream.cuda_stream,
                       ext_obs=buf.data_ptr(), ext_obs_stride=stride) as e:
        print(f"  ext state, stride = n + {pad:<6d} floats ({pad * 4:>7d} B)   {run(e):7.3f} us/step   base {buf.data_ptr():#x}")
    del buf

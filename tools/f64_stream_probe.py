import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda", 0)
n = 1 << 20
for use_stream in (True, False):
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    with pkg.VectorEnv("CartPole-v1", n, seed=1, auto_reset=True, dtype="float64", stream=stream.cuda_stream if use_stream else None) as e:
        for r in (32, 256):
            a = torch.empty((r, n), dtype=torch.int32, device=dev)
            for t in range(r):
                e.SampleActionsDevice(a[t].data_ptr(), seed=2, tick=t)
            e.ResetDevice(); e.RolloutDevice(a.data_ptr(), 128, n, r); e.Sync(); torch.cuda.synchronize()
            ts = []
            for rep in range(6):
                t0 = time.perf_counter(); e.RolloutDevice(a.data_ptr(), 1024, n, r); e.Sync(); torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 1024 * 1e6)
            print("own stream" if not use_stream else "torch stream", "ring", r, e.KernelName(), " ".join(f"{x:.2f}" for x in ts), flush=True)
            del a

"""two_stream_probe: does splitting the 2^20-lane batch into independent halves on separate streams let one half's
write-out overlap the other half's loads / arithmetic?  (Each half is its own handle with its own lane_offset: results
identical to the single batch.)  Wall clock over K steps, launches fed from one host thread per handle."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

pkg = ge.load_package()
dev = torch.device("cuda", 0)
N = 1 << 20
ring, K = 32, 4096
env_name = os.environ.get("ENV", "CartPole-v1")


def run(parts):
    n = N // parts
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    envs, acts = [], []
    for p in range(parts):
        e = pkg.VectorEnv(env_name, n, device=0, seed=3, auto_reset=True, stream=streams[p].cuda_stream, lane_offset=p * n)
        a = torch.empty((ring, n), dtype=torch.int32, device=dev)
        for t in range(ring):
            e.SampleActionsDevice(a[t].data_ptr(), seed=1, tick=t)
        e.ResetDevice()
        e.RolloutDevice(a.data_ptr(), 64, n, ring)
        envs.append(e); acts.append(a)
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        go = threading.Barrier(parts + 1)
        def work(p):
            go.wait()
            envs[p].RolloutDevice(acts[p].data_ptr(), K, n, ring)
        th = [threading.Thread(target=work, args=(p,)) for p in range(parts)]
        for t in th: t.start()
        torch.cuda.synchronize()
        go.wait()
        t0 = time.perf_counter()
        for t in th: t.join()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e6 / K)
    for e in envs: e.Close()
    return sorted(out)[1]


for parts in (1, 2, 4, 1, 2, 4):
    print(f"{env_name}: 2^20 lanes as {parts} independent handle(s) on {parts} stream(s): {run(parts):7.3f} us per step of the whole batch")

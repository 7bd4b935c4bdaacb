"""Measured attainable HBM bandwidth on this box (float4 device-to-device copy), printed as JSON.
DESIGN.md quotes it next to the 8 TB/s spec peak that bench.py's roofline uses."""
import json
import torch

dev = torch.device("cuda", 0)
out = {}
for mib in (32, 256, 2048, 8192):
    n = mib * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(5):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50 if mib <= 2048 else 10
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out[f"copy_{mib}MiB_GBps_read_plus_write"] = 2 * n * 4 / (ms * 1e-3) / 1e9
print(json.dumps(out))

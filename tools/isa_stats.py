#!/usr/bin/env python3
"""Per-kernel statistics of a `hipcc --cuda-device-only -S` listing: instruction counts (total / VALU / packed / VMEM /
s_waitcnt), register use, scratch, occupancy.   python tools/isa_stats.py kernels.s <substring of the mangled name>"""
import re
import sys


def stats(path, pat):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    out = []
    for idx, i in enumerate(starts):
        name = lines[i].split(":")[0]
        if pat not in name:
            continue
        j = starts[idx + 1] if idx + 1 < len(starts) else len(lines)
        seg = lines[i:j]
        code = []
        for l in seg[1:]:
            t = l.strip()
            if t.startswith("s_endpgm"):
                code.append(t)
                break
            if not t or t.startswith((".", ";", "//")) or t.split(";")[0].strip().endswith(":"):
                continue
            code.append(t)
        txt = "\n".join(seg)

        def g(r):
            m = re.search(r, txt)
            return m.group(1) if m else None
        out.append({"name": name, "insts": len(code), "valu": sum(l.startswith("v_") for l in code),
                    "pk": sum(l.startswith("v_pk_") for l in code),
                    "vmem": sum(l.startswith(("global_", "buffer_", "flat_")) for l in code),
                    "lds": sum(l.startswith("ds_") for l in code),
                    "waitcnt": sum(l.startswith("s_waitcnt") for l in code),
                    "vgpr": g(r"; NumVgprs: (\d+)"), "agpr": g(r"; NumAgprs: (\d+)"), "scratch": g(r"; ScratchSize: (\d+)"),
                    "lds_bytes": g(r"; LDSByteSize: (\d+)"), "occupancy": g(r"; Occupancy: (\d+)")})
    return out


if __name__ == "__main__":
    for s in stats(sys.argv[1], sys.argv[2]):
        print(s.pop("name"))
        print("   " + " ".join(f"{k} {v}" for k, v in s.items()))

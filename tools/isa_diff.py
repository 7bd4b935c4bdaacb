#!/usr/bin/env python3
"""Compares the gfx950 instruction streams of two sets of `hipcc --cuda-device-only -S` listings kernel by kernel.
   python tools/isa_diff.py OLD.s[,OLD2.s...] NEW.s[,NEW2.s...]
Kernels are matched by demangled name with the parameter list dropped (so a change of an argument struct's NAME does not hide
an unchanged kernel); prints identical / changed / only-in-one-side counts and, for changed kernels, the instruction counts."""
import re
import subprocess
import sys


def kernels(paths):
    out = {}
    for path in paths.split(","):
        lines = open(path).read().split("\n")
        starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
        for idx, i in enumerate(starts):
            name = lines[i].split(":")[0]
            j = starts[idx + 1] if idx + 1 < len(starts) else len(lines)
            code = []
            for l in lines[i + 1:j]:
                t = l.split(";")[0].strip()
                if t.startswith("s_endpgm"):
                    code.append(t)
                    break
                if not t or t.startswith((".", "//")) or t.endswith(":"):
                    continue
                code.append(re.sub(r"\.LBB\d+_", ".LBB_", re.sub(r"\s+", " ", t)))
            out[name] = code
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    res = {}
    for m, d in zip(names, dem):
        d = re.sub(r"\(.*\)$", "", d.replace("void ", "", 1)).strip()
        res[d] = out[m]
    return res


if __name__ == "__main__":
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    same = [k for k in a if k in b and a[k] == b[k]]
    diff = [k for k in a if k in b and a[k] != b[k]]
    print(f"identical {len(same)}  changed {len(diff)}  only-old {len(set(a) - set(b))}  only-new {len(set(b) - set(a))}")
    for k in diff:
        print(f"  CHANGED {k}: {len(a[k])} -> {len(b[k])} instructions")
    if "-v" in sys.argv:
        for k in sorted(set(a) - set(b)):
            print("  only-old", k)
        for k in sorted(set(b) - set(a)):
            print("  only-new", k)

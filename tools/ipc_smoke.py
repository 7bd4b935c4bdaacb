#!/usr/bin/env python3
"""Does HIP IPC (hipIpcGetMemHandle / hipIpcOpenMemHandle) work between two processes on this box?  Parent allocates and
fills a buffer, child opens the handle, reads it back and writes a marker the parent then sees."""
import ctypes as C
import multiprocessing as mp
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def hip():
    import importlib.util
    spec = importlib.util.find_spec("torch")
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    return C.CDLL(p if os.path.exists(p) else "libamdhip64.so")


class Handle(C.Structure):
    _fields_ = [("reserved", C.c_char * 64)]


def chk(rc, what):
    assert rc == 0, f"{what} -> {rc}"


def child(handle_bytes, q):
    h = hip()
    chk(h.hipSetDevice(0), "set")
    handle = Handle.from_buffer_copy(handle_bytes)
    ptr = C.c_void_p()
    h.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]          # the handle travels BY VALUE
    rc = h.hipIpcOpenMemHandle(C.byref(ptr), handle, C.c_uint(1))
    if rc != 0:
        q.put(("open failed", rc)); return
    buf = (C.c_float * 16)()
    chk(h.hipMemcpy(buf, ptr, 64, 2), "d2h")
    vals = list(buf)
    buf[0] = 777.0
    chk(h.hipMemcpy(ptr, buf, 4, 1), "h2d")
    chk(h.hipDeviceSynchronize(), "sync")
    chk(h.hipIpcCloseMemHandle(ptr), "close")
    q.put(("ok", vals[:4]))


if __name__ == "__main__":
    mp.set_start_method("spawn")
    h = hip()
    chk(h.hipSetDevice(0), "set")
    ptr = C.c_void_p()
    chk(h.hipMalloc(C.byref(ptr), 1 << 20), "malloc")
    src = (C.c_float * 16)(*[float(i + 1) for i in range(16)])
    chk(h.hipMemcpy(ptr, src, 64, 1), "h2d")
    handle = Handle()
    rc = h.hipIpcGetMemHandle(C.byref(handle), ptr)
    print("hipIpcGetMemHandle ->", rc)
    if rc == 0:
        q = mp.Queue()
        p = mp.Process(target=child, args=(bytes(handle), q))
        p.start()
        print("child:", q.get(timeout=60))
        p.join()
        back = (C.c_float * 4)()
        chk(h.hipMemcpy(back, ptr, 16, 2), "d2h")
        print("parent sees", list(back))

"""oracle/capi.py — ctypes loader for oracle/build/liboracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY.  Importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
leg.  The product package (gym.net_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "liboracle.so")
_lib = None

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the C restatement with oracle/Makefile (gcc).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("classic_control_ref.c", "cpu_baseline.c", "Makefile")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    L.ref_cartpole_constants.argtypes = [_f64p]
    L.ref_cartpole_step_f64.argtypes = [_f64p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L.ref_cartpole_step_f64.restype = C.c_int
    L.ref_cartpole_step_f32.argtypes = [_f32p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L.ref_cartpole_step_f32.restype = C.c_int
    L.ref_cartpole_step_batch_f64.argtypes = [_f64p, _i32p, _i32p, _f32p, _u8p, C.c_int64]
    L.ref_cartpole_step_batch_f32.argtypes = [_f32p, _i32p, _i32p, _f32p, _u8p, C.c_int64]
    L.ref_sincos_f32_kernel.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ref_div_total_mass_kernel.argtypes = [C.c_float]
    L.ref_div_total_mass_kernel.restype = C.c_float
    L.ref_check_div_total_mass.argtypes = [_i32p, C.c_int32]
    L.ref_check_div_total_mass.restype = C.c_int64
    L.ref_discrete_contains.argtypes = [C.c_int, C.c_int]
    L.ref_discrete_contains.restype = C.c_int
    L.ref_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
    L.ref_reset_words.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _u32p]
    L.ref_cartpole_reset_batch_f32.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _f32p, C.c_int64]
    L.ref_discrete_sample_batch.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, _i32p, C.c_int64]
    L.ref_discrete_sample_masked_batch.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, _u8p, C.c_int64, _i32p, C.c_int64]
    L.ref_compose_discrete_batch.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_float, _i32p, _i32p, C.c_int64]
    L.ref_box_uniform_sample_batch.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float, _f32p, C.c_int64]
    L.ref_pendulum_step_f64.argtypes = [_f64p, C.c_double, _f64p, C.POINTER(C.c_double)]
    L.ref_pendulum_step_f32.argtypes = [_f32p, C.c_float, _f32p, C.POINTER(C.c_float)]
    L.ref_pendulum_reset_f32.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _f32p]
    L.ref_mountaincar_step_f64.argtypes = [_f64p, C.c_int, C.POINTER(C.c_double)]
    L.ref_mountaincar_step_f64.restype = C.c_int
    L.ref_mountaincar_step_f32.argtypes = [_f32p, C.c_int, C.POINTER(C.c_float)]
    L.ref_mountaincar_step_f32.restype = C.c_int
    L.ref_mountaincar_reset_f32.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _f32p]
    L.ref_acrobot_step_f64.argtypes = [_f64p, C.c_int, _f64p, C.POINTER(C.c_double)]
    L.ref_acrobot_step_f64.restype = C.c_int
    L.ref_acrobot_step_f32.argtypes = [_f32p, C.c_int, _f32p, C.POINTER(C.c_float)]
    L.ref_acrobot_step_f32.restype = C.c_int
    L.ref_acrobot_reset_f32.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _f32p]
    L.ref_cpu_baseline_run.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_uint64,
                                       C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.ref_cpu_baseline_run.restype = C.c_double
    _lib = L
    return L


# ---------------------------------------------------------------------------------------------
# NumPy-friendly wrappers.  State arrays are structure-of-arrays [dim, n], as in the engine.
# ---------------------------------------------------------------------------------------------
def cartpole_constants():
    out = np.zeros(10, dtype=np.float64)
    lib().ref_cartpole_constants(out)
    names = ("gravity", "masscart", "masspole", "total_mass", "length", "polemass_length",
             "force_mag", "tau", "theta_threshold_radians", "x_threshold")
    return dict(zip(names, out.tolist()))


def cartpole_step(state, action, sbd=None, dtype=np.float64):
    """Batched CartPoleEnv.Step.  Returns (new_state[4,n] dtype, reward f32[n], done u8[n], sbd i32[n])."""
    s = np.ascontiguousarray(np.array(state, dtype=dtype, copy=True))
    n = s.shape[1]
    a = np.ascontiguousarray(np.asarray(action, dtype=np.int32))
    b = np.full(n, -1, dtype=np.int32) if sbd is None else np.ascontiguousarray(np.array(sbd, dtype=np.int32, copy=True))
    reward = np.zeros(n, dtype=np.float32)
    done = np.zeros(n, dtype=np.uint8)
    if dtype == np.float64:
        lib().ref_cartpole_step_batch_f64(s, a, b, reward, done, n)
    else:
        lib().ref_cartpole_step_batch_f32(s, a, b, reward, done, n)
    return s, reward, done, b


def sincos_kernel(x):
    """The kernels' sin/cos restated on the CPU (bit-identical to the GPU for |x| <= 65536)."""
    x = np.asarray(x, dtype=np.float32).reshape(-1)
    s = np.empty_like(x); c = np.empty_like(x)
    fs, fc = C.c_float(), C.c_float()
    L = lib()
    for i, v in enumerate(x):
        L.ref_sincos_f32_kernel(float(v), C.byref(fs), C.byref(fc))
        s[i], c[i] = fs.value, fc.value
    return s, c


def check_div_total_mass(biased_exponents):
    e = np.ascontiguousarray(np.asarray(biased_exponents, dtype=np.int32))
    return int(lib().ref_check_div_total_mass(e, e.shape[0]))


def philox4x32_10(ctr, key):
    out = np.zeros(4, dtype=np.uint32)
    lib().ref_philox4x32_10(np.asarray(ctr, dtype=np.uint32), np.asarray(key, dtype=np.uint32), out)
    return out


def cartpole_reset(seed, lane0, tick, n):
    out = np.zeros((4, n), dtype=np.float32)
    lib().ref_cartpole_reset_batch_f32(seed, lane0, tick, out, n)
    return out


def discrete_sample(seed, lane0, tick, nvals, start, count):
    out = np.zeros(count, dtype=np.int32)
    lib().ref_discrete_sample_batch(seed, lane0, tick, nvals, start, out, count)
    return out


def discrete_sample_masked(seed, lane0, tick, nvals, start, mask, count):
    """Discrete.Sample(mask): mask uint8 [count, nvals] (per lane) or [nvals] (shared row)."""
    mask = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
    stride = nvals if mask.ndim == 2 else 0
    out = np.zeros(count, dtype=np.int32)
    lib().ref_discrete_sample_masked_batch(seed, lane0, tick, nvals, start, mask.reshape(-1), stride, out, count)
    return out


def compose_discrete(seed, lane0, tick, nvals, epsilon, policy):
    policy = np.ascontiguousarray(np.asarray(policy, dtype=np.int32))
    out = np.zeros_like(policy)
    lib().ref_compose_discrete_batch(seed, lane0, tick, nvals, epsilon, policy, out, policy.shape[0])
    return out


def box_uniform_sample(seed, lane0, tick, low, high, count):
    out = np.zeros(count, dtype=np.float32)
    lib().ref_box_uniform_sample_batch(seed, lane0, tick, low, high, out, count)
    return out


def _per_lane(fn_reset, dim, seed, lane0, tick, n):
    out = np.zeros((dim, n), dtype=np.float32)
    tmp = np.zeros(dim, dtype=np.float32)
    for i in range(n):
        fn_reset(seed, lane0 + i, tick, tmp)
        out[:, i] = tmp
    return out


def pendulum_reset(seed, lane0, tick, n):
    return _per_lane(lib().ref_pendulum_reset_f32, 2, seed, lane0, tick, n)


def mountaincar_reset(seed, lane0, tick, n):
    return _per_lane(lib().ref_mountaincar_reset_f32, 2, seed, lane0, tick, n)


def acrobot_reset(seed, lane0, tick, n):
    return _per_lane(lib().ref_acrobot_reset_f32, 4, seed, lane0, tick, n)


def pendulum_step(state, action, dtype=np.float64):
    """Returns (new_state[2,n], obs[3,n], reward[n], done u8[n] (always 0))."""
    s = np.array(state, dtype=dtype, copy=True)
    n = s.shape[1]
    obs = np.zeros((3, n), dtype=dtype)
    rew = np.zeros(n, dtype=dtype)
    L = lib()
    st = np.zeros(2, dtype=dtype); o = np.zeros(3, dtype=dtype)
    for i in range(n):
        st[:] = s[:, i]
        if dtype == np.float64:
            r = C.c_double(); L.ref_pendulum_step_f64(st, float(action[i]), o, C.byref(r))
        else:
            r = C.c_float(); L.ref_pendulum_step_f32(st, float(action[i]), o, C.byref(r))
        s[:, i] = st; obs[:, i] = o; rew[i] = r.value
    return s, obs, rew, np.zeros(n, dtype=np.uint8)


def mountaincar_step(state, action, dtype=np.float64):
    """Returns (new_state[2,n], reward[n], done u8[n])."""
    s = np.array(state, dtype=dtype, copy=True)
    n = s.shape[1]
    rew = np.zeros(n, dtype=dtype); done = np.zeros(n, dtype=np.uint8)
    L = lib()
    st = np.zeros(2, dtype=dtype)
    for i in range(n):
        st[:] = s[:, i]
        if dtype == np.float64:
            r = C.c_double(); d = L.ref_mountaincar_step_f64(st, int(action[i]), C.byref(r))
        else:
            r = C.c_float(); d = L.ref_mountaincar_step_f32(st, int(action[i]), C.byref(r))
        s[:, i] = st; rew[i] = r.value; done[i] = d
    return s, rew, done


def acrobot_step(state, action, dtype=np.float64):
    """Returns (new_state[4,n], obs[6,n], reward[n], done u8[n])."""
    s = np.array(state, dtype=dtype, copy=True)
    n = s.shape[1]
    obs = np.zeros((6, n), dtype=dtype)
    rew = np.zeros(n, dtype=dtype); done = np.zeros(n, dtype=np.uint8)
    L = lib()
    st = np.zeros(4, dtype=dtype); o = np.zeros(6, dtype=dtype)
    for i in range(n):
        st[:] = s[:, i]
        if dtype == np.float64:
            r = C.c_double(); d = L.ref_acrobot_step_f64(st, int(action[i]), o, C.byref(r))
        else:
            r = C.c_float(); d = L.ref_acrobot_step_f32(st, int(action[i]), o, C.byref(r))
        s[:, i] = st; obs[:, i] = o; rew[i] = r.value; done[i] = d
    return s, obs, rew, done


def cpu_baseline(n_envs, steps, threads, alloc_faithful=True, seed=0x5EED):
    """Times the per-instance f64 CartPole path on host cores.  Returns dict(seconds, env_steps, ...)."""
    es = C.c_int64(); cs = C.c_double(); dn = C.c_int64()
    sec = lib().ref_cpu_baseline_run(n_envs, steps, threads, 1 if alloc_faithful else 0, seed,
                                     C.byref(es), C.byref(cs), C.byref(dn))
    return {"seconds": sec, "env_steps": es.value, "checksum": cs.value, "dones": dn.value,
            "steps_per_sec": es.value / sec if sec > 0 else float("nan")}

"""oracle/capi.py — ctypes loader for oracle/build/liboracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY.  Importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
leg.  The product package (gym.net_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GYMNET_ORACLE_SO: load another build of the same restatement (the sanitizer builds of oracle/Makefile; tests/test_sanitizers.py)
_SO = os.environ.get("GYMNET_ORACLE_SO") or os.path.join(_HERE, "build", "liboracle.so")
_lib = None

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the C restatement with oracle/Makefile (gcc).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("classic_control_ref.c", "cpu_baseline.c", "Makefile")]
    if os.environ.get("GYMNET_ORACLE_SO"):
        return _SO                       # a sanitizer build made by `make -C oracle asan|ubsan|tsan`
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
    L.ref_sincos_f32_small.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ref_div_total_mass_kernel.argtypes = [C.c_float]
    L.ref_div_total_mass_kernel.restype = C.c_float
    L.ref_check_div_total_mass.argtypes = [_i32p, C.c_int32]
    L.ref_check_div_total_mass.restype = C.c_int64
    L.ref_fmod_2pi_kernel.argtypes = [C.c_float]
    L.ref_fmod_2pi_kernel.restype = C.c_float
    L.ref_check_fmod_2pi.argtypes = [C.c_uint32, C.c_int32]
    L.ref_check_fmod_2pi.restype = C.c_int64
    L.ref_discrete_contains.argtypes = [C.c_int, C.c_int]
    L.ref_discrete_contains.restype = C.c_int
    L.ref_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
    L.ref_reset_words.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _u32p]
    L.ref_cartpole_reset_batch_f32.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _f32p, C.c_int64]
    L.ref_action_words_batch.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _u32p, _u32p, C.c_int64]
    L.ref_discrete_sample_batch.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, _i32p, C.c_int64]
    L.ref_discrete_sample_masked_batch.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, _u8p, C.c_int64, _i32p, C.c_int64]
    L.ref_compose_discrete_batch.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_float, _i32p, _i32p, C.c_int64]
    L.ref_box_uniform_sample_batch.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float, _f32p, C.c_int64]
    L.ref_div_total_mass_f64_kernel.argtypes = [C.c_double]
    L.ref_div_total_mass_f64_kernel.restype = C.c_double
    L.ref_check_div_total_mass_f64.argtypes = [C.c_uint64, C.c_int64]
    L.ref_check_div_total_mass_f64.restype = C.c_int64
    L.ref_sincos_f64_kernel_batch.argtypes = [_f64p, _f64p, _f64p, C.c_int64]
    L.ref_cartpole_step_batch_f64_kernel.argtypes = [_f64p, _i32p, _i32p, _f32p, _u8p, C.c_int64]
    L.ref_cartpole_reset_batch_f64.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, _f64p, C.c_int64]
    L.ref_cartpole_autoreset_step_batch_f64.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, _f64p, _i32p, _f32p, _u8p, C.c_int64]
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
    L.ref_acrobot_step_batch_f32_literal.argtypes = [_f32p, _i32p, _f32p, _f32p, _u8p, C.c_int64]
    L.ref_acrobot_reset_f32.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _f32p]
    L.ref_env_step_batch_f32.argtypes = [C.c_int, _f32p, C.c_void_p, C.c_void_p, _f32p, _f32p, _u8p, C.c_int64]
    L.ref_env_step_batch_f64.argtypes = [C.c_int, _f64p, C.c_void_p, C.c_void_p, _f64p, _f64p, _u8p, C.c_int64]
    L.ref_env_reset_batch_f32.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, _f32p, C.c_void_p, C.c_int64]
    L.ref_env_autoreset_step_batch_f32.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, _f32p, C.c_void_p,
                                                   _f32p, _f32p, _u8p, C.c_int64]
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


def cartpole_step(state, action, sbd=None, dtype=np.float64, kernel_sincos=False):
    """Batched CartPoleEnv.Step.  Returns (new_state[4,n] dtype, reward f32[n], done u8[n], sbd i32[n]).
    dtype float64: the reference's arithmetic with libm sin / cos; kernel_sincos=True swaps in the GYMNET_FLAG_F64 kernel's own
    sin / cos (the bit-identical twin of the float64 HIP kernel).  dtype float32: the float32 kernel's twin."""
    s = np.ascontiguousarray(np.array(state, dtype=dtype, copy=True))
    n = s.shape[1]
    a = np.ascontiguousarray(np.asarray(action, dtype=np.int32))
    b = np.full(n, -1, dtype=np.int32) if sbd is None else np.ascontiguousarray(np.array(sbd, dtype=np.int32, copy=True))
    reward = np.zeros(n, dtype=np.float32)
    done = np.zeros(n, dtype=np.uint8)
    if dtype == np.float64:
        (lib().ref_cartpole_step_batch_f64_kernel if kernel_sincos else lib().ref_cartpole_step_batch_f64)(s, a, b, reward, done, n)
    else:
        lib().ref_cartpole_step_batch_f32(s, a, b, reward, done, n)
    return s, reward, done, b


def sincos_f64_kernel(x):
    """The GYMNET_FLAG_F64 kernel's float64 sin/cos restated on the CPU (bit-identical to the GPU for |x| <= 823549)."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    s = np.empty_like(x); c = np.empty_like(x)
    lib().ref_sincos_f64_kernel_batch(x, s, c, x.shape[0])
    return s, c


def _lane_seed_ptr(lane_seed):
    if lane_seed is None:
        return None, None
    keep = np.ascontiguousarray(np.asarray(lane_seed, dtype=np.uint64))
    return keep, keep.ctypes.data_as(C.c_void_p)


def cartpole_reset_f64(seed, lane0, tick, n, lane_seed=None):
    """Reset draw of a GYMNET_FLAG_F64 handle: float64 SoA [4, n], 53-bit uniforms from two Philox calls per lane."""
    out = np.zeros((4, n), dtype=np.float64)
    keep, ls = _lane_seed_ptr(lane_seed)
    lib().ref_cartpole_reset_batch_f64(seed, ls, lane0, tick, out, n)
    return out


def cartpole_autoreset_step_f64(seed, lane0, tick, state, action, lane_seed=None):
    """One vector step of a GYMNET_FLAG_F64 handle with the fused auto-reset.  Returns (state[4,n] f64, reward, done u8)."""
    s = np.ascontiguousarray(np.array(state, dtype=np.float64, copy=True))
    n = s.shape[1]
    a = np.ascontiguousarray(np.asarray(action, dtype=np.int32))
    rew = np.zeros(n, dtype=np.float32); done = np.zeros(n, dtype=np.uint8)
    keep, ls = _lane_seed_ptr(lane_seed)
    lib().ref_cartpole_autoreset_step_batch_f64(seed, ls, lane0, tick, s, a, rew, done, n)
    return s, rew, done


def sincos_kernel(x, small=False):
    """The kernels' sin/cos restated on the CPU (bit-identical to the GPU for |x| <= 65536); small=True: the
    two-constant small-argument form of the Acrobot kernel (|x| < 24)."""
    x = np.asarray(x, dtype=np.float32).reshape(-1)
    s = np.empty_like(x); c = np.empty_like(x)
    fs, fc = C.c_float(), C.c_float()
    L = lib()
    fn = L.ref_sincos_f32_small if small else L.ref_sincos_f32_kernel
    for i, v in enumerate(x):
        fn(float(v), C.byref(fs), C.byref(fc))
        s[i], c[i] = fs.value, fc.value
    return s, c


def check_div_total_mass(biased_exponents):
    e = np.ascontiguousarray(np.asarray(biased_exponents, dtype=np.int32))
    return int(lib().ref_check_div_total_mass(e, e.shape[0]))


def check_div_total_mass_f64(seed, count):
    """Number of sampled binary64 dividends for which the float64 kernel's fma-pair constant division differs from IEEE x / total_mass
    (must be 0; the proof is tools/prove_div_total_mass_f64.py)."""
    return int(lib().ref_check_div_total_mass_f64(seed, count))


def check_fmod_2pi(stride, multiples):
    """Number of arguments for which the Pendulum kernel's fast fmod(., 2 pi) differs from libm's fmodf (must be 0)."""
    return int(lib().ref_check_fmod_2pi(stride, multiples))


def philox4x32_10(ctr, key):
    out = np.zeros(4, dtype=np.uint32)
    lib().ref_philox4x32_10(np.asarray(ctr, dtype=np.uint32), np.asarray(key, dtype=np.uint32), out)
    return out


def cartpole_reset(seed, lane0, tick, n):
    out = np.zeros((4, n), dtype=np.float32)
    lib().ref_cartpole_reset_batch_f32(seed, lane0, tick, out, n)
    return out


def action_words(seed, lane0, tick, count):
    """Words A and B of the action stream (version 2: one Philox call per group of four global lanes) for `count` lanes."""
    a, b = np.zeros(count, dtype=np.uint32), np.zeros(count, dtype=np.uint32)
    lib().ref_action_words_batch(seed, lane0, tick, a, b, count)
    return a, b


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


ENV_IDS = {"CartPole-v1": 0, "Pendulum-v1": 1, "MountainCar-v0": 2, "Acrobot-v1": 3}
_DIMS = {0: (4, 4), 1: (2, 3), 2: (2, 2), 3: (4, 6)}          # env_id -> (state dim, observation dim)


def _env_id(env):
    return ENV_IDS[env] if isinstance(env, str) else int(env)


def _actions(env_id, action):
    return np.ascontiguousarray(np.asarray(action, dtype=np.float32 if env_id == 1 else np.int32))


def env_step(env, state, action, sbd=None, dtype=np.float64):
    """One batched step of any env over SoA state [S, n] (loops the per-instance restatement in C).
    Returns (new_state[S,n], obs[O,n], reward[n], done u8[n]) in `dtype`; sbd (CartPole, int32[n]) is updated in place."""
    e = _env_id(env)
    S, O = _DIMS[e]
    s = np.ascontiguousarray(np.array(state, dtype=dtype, copy=True))
    assert s.shape[0] == S
    n = s.shape[1]
    a = _actions(e, action)
    obs = np.zeros((O, n), dtype=dtype); rew = np.zeros(n, dtype=dtype); done = np.zeros(n, dtype=np.uint8)
    b = None if sbd is None else sbd.ctypes.data_as(C.c_void_p)
    fn = lib().ref_env_step_batch_f64 if dtype == np.float64 else lib().ref_env_step_batch_f32
    fn(e, s, a.ctypes.data_as(C.c_void_p), b, obs, rew, done, n)
    return s, obs, rew, done


def env_reset(env, seed, lane0, tick, n, with_obs=False):
    """The engine's reset draw for global lanes [lane0, lane0 + n) at `tick` (float32 SoA [S, n]); with_obs adds the observation."""
    e = _env_id(env)
    S, O = _DIMS[e]
    s = np.zeros((S, n), dtype=np.float32)
    o = np.zeros((O, n), dtype=np.float32) if with_obs else None
    lib().ref_env_reset_batch_f32(e, seed, lane0, tick, s, None if o is None else o.ctypes.data_as(C.c_void_p), n)
    return (s, o) if with_obs else s


def env_autoreset_step(env, seed, lane0, tick, state, action, lane_seed=None):
    """One vector step with the fused auto-reset, kernel (float32) semantics: finished lanes are re-drawn with
    Philox(seed or lane_seed[i], (lane0 + i, tick)).  Returns (state[S,n], obs[O,n], reward[n], done u8[n])."""
    e = _env_id(env)
    S, O = _DIMS[e]
    s = np.ascontiguousarray(np.array(state, dtype=np.float32, copy=True))
    n = s.shape[1]
    a = _actions(e, action)
    obs = np.zeros((O, n), dtype=np.float32); rew = np.zeros(n, dtype=np.float32); done = np.zeros(n, dtype=np.uint8)
    ls = None
    if lane_seed is not None:
        lane_seed = np.ascontiguousarray(np.asarray(lane_seed, dtype=np.uint64))
        ls = lane_seed.ctypes.data_as(C.c_void_p)
    lib().ref_env_autoreset_step_batch_f32(e, seed, ls, lane0, tick, s, a.ctypes.data_as(C.c_void_p), obs, rew, done, n)
    return s, obs, rew, done


def pendulum_reset(seed, lane0, tick, n):
    return env_reset(1, seed, lane0, tick, n)


def mountaincar_reset(seed, lane0, tick, n):
    return env_reset(2, seed, lane0, tick, n)


def acrobot_reset(seed, lane0, tick, n):
    return env_reset(3, seed, lane0, tick, n)


def pendulum_step(state, action, dtype=np.float64):
    """Returns (new_state[2,n], obs[3,n], reward[n], done u8[n] (always 0))."""
    return env_step(1, state, action, dtype=dtype)


def mountaincar_step(state, action, dtype=np.float64):
    """Returns (new_state[2,n], reward[n], done u8[n])."""
    s, _, r, d = env_step(2, state, action, dtype=dtype)
    return s, r, d


def acrobot_step(state, action, dtype=np.float64):
    """Returns (new_state[4,n], obs[6,n], reward[n], done u8[n])."""
    return env_step(3, state, action, dtype=dtype)


def acrobot_step_f32_literal(state, action):
    """One Acrobot step in a LITERAL float32 transcription of upstream's formulas (libm sinf / cosf, IEEE division, upstream's
    association) — not what the kernel runs; the yardstick of tools/acrobot_accuracy.py.  Returns (state[4,n], obs[6,n], reward, done)."""
    s = np.ascontiguousarray(np.array(state, dtype=np.float32, copy=True))
    n = s.shape[1]
    a = np.ascontiguousarray(np.asarray(action, dtype=np.int32))
    obs = np.zeros((6, n), np.float32); rew = np.zeros(n, np.float32); done = np.zeros(n, np.uint8)
    lib().ref_acrobot_step_batch_f32_literal(s, a, obs, rew, done, n)
    return s, obs, rew, done


def cpu_baseline(n_envs, steps, threads, alloc_faithful=True, seed=0x5EED):
    """Times the per-instance f64 CartPole path on host cores.  Returns dict(seconds, env_steps, ...)."""
    es = C.c_int64(); cs = C.c_double(); dn = C.c_int64()
    sec = lib().ref_cpu_baseline_run(n_envs, steps, threads, 1 if alloc_faithful else 0, seed,
                                     C.byref(es), C.byref(cs), C.byref(dn))
    return {"seconds": sec, "env_steps": es.value, "checksum": cs.value, "dones": dn.value,
            "steps_per_sec": es.value / sec if sec > 0 else float("nan")}

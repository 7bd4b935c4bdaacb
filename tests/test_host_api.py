"""CPU tests: the host-side mirror of the reference interface (spaces, Step, errors), and that the C-ABI
library loads and exports every symbol include/gymnet_amd.h declares.  No compute calls: there is no GPU
here and the engine has no CPU fallback — which is itself asserted."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_all_exported(gymnet):
    hdr = open(os.path.join(ROOT, "include", "gymnet_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gymnet_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 35
    lib = ctypes.CDLL(gymnet.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    # and the ctypes binding covers exactly the declared set
    assert declared == set(gymnet._capi.PROTOTYPES)
    assert lib.gymnet_abi_version() == 1


def test_struct_layouts_match_the_header(gymnet):
    c = gymnet._capi
    assert ctypes.sizeof(c.Config) == 72 and c.Config.seed.offset == 32 and c.Config.max_episode_steps.offset == 64
    assert ctypes.sizeof(c.EnvInfo) == 144 and c.EnvInfo.obs_low.offset == 68
    assert ctypes.sizeof(c.Counters) == 40
    assert ctypes.sizeof(c.DeviceView) == 40 + 11 * 8 + 8


def test_env_descriptions_match_the_reference_ctor(gymnet):
    i = gymnet.env_describe(0)                                   # CartPoleEnv.cs:43-52
    assert i.name == b"CartPole-v1" and i.obs_dim == 4 and i.state_dim == 4 and i.obs_aliases_state == 1
    assert i.action_is_box == 0 and i.action_n == 2              # Discrete(2)
    high = np.array(i.obs_high[:4], dtype=np.float32)
    assert high[0] == np.float32(2.4) * 2 and high[1] == np.finfo(np.float32).max
    assert high[2] == np.float32(12 * 2 * np.pi / 360) * 2 and high[3] == np.finfo(np.float32).max
    assert np.array_equal(np.array(i.obs_low[:4], dtype=np.float32), -high)
    assert i.algorithmic_bytes_per_step == 41
    assert [gymnet.env_describe(k).algorithmic_bytes_per_step for k in (1, 2, 3)] == [37, 25, 65]
    assert gymnet.env_describe(1).action_is_box == 1 and gymnet.env_describe(3).obs_dim == 6
    with pytest.raises(ValueError):
        gymnet.env_describe(9)


def test_no_gpu_means_loud_failure_not_fallback(gymnet):
    if os.path.exists("/dev/kfd") and gymnet.device_count() > 0:
        pytest.skip("a GPU is present")
    assert gymnet.device_count() == 0
    with pytest.raises(gymnet.NoDeviceError):
        gymnet.VectorEnv("CartPole-v1", 8)
    with pytest.raises(gymnet.NoDeviceError):
        gymnet.CartPoleEnv()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gym.net_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".cs")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "liboracle" not in text and "classic_control_ref" not in text, f


# ---- Box / Discrete: tests/Gym.Tests/Spaces/BoxTest.cs:14-42 and Discrete.cs:38-40 -----------------
def test_box_bounded(gymnet):
    Box = gymnet.Box
    box = Box(-5.0, 5.0, dtype=np.float32)
    assert box.IsBounded(Box.BOTH)
    box = Box(-np.inf, 5.0, dtype=np.float32)
    assert not box.IsBounded(Box.BELOW) and box.IsBounded(Box.ABOVE) and not box.IsBounded(Box.BOTH)
    box = Box(5.0, np.inf, dtype=np.float32)
    assert not box.IsBounded(Box.ABOVE) and box.IsBounded(Box.BELOW) and not box.IsBounded(Box.BOTH)
    box = Box(-np.inf, np.inf, dtype=np.float32)
    assert not box.IsBounded(Box.ABOVE) and not box.IsBounded(Box.BELOW) and not box.IsBounded(Box.BOTH)


def test_box_bounded_sampling(gymnet):
    box = gymnet.Box(-5.0, 5.0, seed=3)
    for _ in range(100):
        s = float(box.Sample(None))
        assert -5.0 <= s <= 5.0
    with pytest.raises(NotImplementedError):
        box.Sample(mask=np.ones(1))
    # the reference's one-sided regimes (Box.cs:83-84): exponential(1) + bound
    lo = gymnet.Box(np.array([2.0, -np.inf]), np.array([np.inf, 7.0]), seed=1)
    s = np.array([lo.Sample() for _ in range(200)])
    assert (s[:, 0] >= 2.0).all() and (s[:, 1] >= 7.0).all()
    assert lo.Contains(np.array([3.0, 1.0], dtype=np.float32)) and not lo.Contains(np.array([1.0, 1.0], dtype=np.float32))
    assert gymnet.Box(-1.0, 1.0, (2,)) == gymnet.Box(-1.0, 1.0, (2,))
    with pytest.raises(NotImplementedError):
        lo.Contains([3.0, 1.0])


def test_discrete(gymnet):
    d = gymnet.Discrete(2, seed=5)
    assert d.Contains(0) and d.Contains(1) and not d.Contains(2) and not d.Contains(-1)    # 0 <= x < N
    assert {d.Sample() for _ in range(64)} == {0, 1}
    assert gymnet.Discrete(3, start=10, seed=1).Sample() in (10, 11, 12)
    assert gymnet.Discrete(3, start=10, seed=1).Sample(mask=np.array([0, 0, 1])) == 12
    assert gymnet.Discrete(3, start=10, seed=1).Sample(mask=np.array([0, 0, 0])) == 10
    with pytest.raises(NotImplementedError):
        d.Contains("x")
    assert repr(d) == "Discrete(2)" and d.Shape == (2,)
    import enum

    class Push(enum.Enum):                                          # Contains(Enum), Discrete.cs:42-44
        Left = 0
        Right = 1
        Up = 2
    assert d.Contains(Push.Left) and d.Contains(Push.Right) and not d.Contains(Push.Up)


def test_step_record(gymnet):
    Step = gymnet.Step
    s = Step(np.arange(4.0), 1.0, False, None)
    observation, reward, done, information = s                       # Deconstruct, Step.cs:24-29
    assert reward == 1.0 and done is False and information is None and observation[3] == 3.0
    assert s == Step(np.arange(4.0), 1.0, False, None) and s != Step(np.arange(4.0), 0.0, False, None)
    c = s.Clone()
    c.Observation[0] = 9
    assert s.Observation[0] == 0 and "Reward: 1.0" in repr(s)
    b = gymnet.BatchStep(np.zeros((3, 4), np.float32), np.ones(3, np.float32), np.array([0, 1, 0], bool))
    assert len(b) == 3 and b[1].Done is True and [x.Reward for x in b] == [1.0, 1.0, 1.0]


def test_error_vocabulary(gymnet):
    assert str(gymnet.InvalidActionError()) == "Action is outside of the configured action space."
    assert str(gymnet.AlreadySteppingError()) == "already running an async step"
    assert str(gymnet.NotSteppingError()) == "not running an async step"
    c = gymnet._capi
    assert c.load_library().gymnet_status_string(c.ERR_INVALID_ACTION) == b"Action is outside of the configured action space."


def test_shard_plan(gymnet):
    p = gymnet.ShardPlan(1 << 23, 8)
    assert p.even and [p.shard(r) for r in (0, 7)] == [(0, 1 << 20), (7 << 20, 1 << 20)]
    q = gymnet.ShardPlan(10, 3)
    assert [q.shard(r) for r in range(3)] == [(0, 3), (3, 3), (6, 4)] and not q.even
    assert sum(q.count(r) for r in range(3)) == 10
    for lane in range(10):
        r, i = q.owner(lane)
        assert q.offset(r) + i == lane and 0 <= i < q.count(r)
    with pytest.raises(ValueError):
        gymnet.ShardPlan(2, 3)


def test_missing_extension_raises_instead_of_falling_back(tmp_path):
    # a fresh interpreter whose package copy has no lib/: loading must raise, and nothing CPU-side may take over
    import shutil
    import subprocess
    import sys
    pkg_copy = tmp_path / "gym.net_amd"
    shutil.copytree(os.path.join(ROOT, "gym.net_amd"), pkg_copy, ignore=shutil.ignore_patterns("lib", "__pycache__", "csrc", "csharp"))
    code = (
        "import importlib.util, sys\n"
        f"spec = importlib.util.spec_from_file_location('gymnet_amd', r'{pkg_copy}/__init__.py', submodule_search_locations=[r'{pkg_copy}'])\n"
        "m = importlib.util.module_from_spec(spec); sys.modules['gymnet_amd'] = m; spec.loader.exec_module(m)\n"
        "try:\n    m.VectorEnv('CartPole-v1', 4)\nexcept m.GymNetError as e:\n    print('RAISED', e)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "RAISED" in r.stdout and "has not been built" in r.stdout and "no CPU fallback" in r.stdout, r.stdout + r.stderr


def test_csharp_binding_sources_lex_cleanly_and_cover_the_header():
    """No .NET toolchain exists here, so the C# binding cannot be compiled; the least that can be checked is that
    both files tokenize without a single error token (pygments' C# lexer), that braces / parentheses balance, and
    that Native.cs declares a [DllImport] for every function the header exports (minus none)."""
    from pygments.lexers.dotnet import CSharpLexer
    from pygments.token import Error
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gymnet_amd.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(gymnet_[a-z_0-9]+)\s*\(", hdr))
    srcs = {}
    for f in ("Native.cs", "VectorEnv.cs"):
        text = open(os.path.join(ROOT, "gym.net_amd", "csharp", f)).read()
        srcs[f] = text
        toks = list(CSharpLexer().get_tokens(text))
        assert not [v for tt, v in toks if tt is Error], f
        code = re.sub(r"//.*", "", text)
        code = re.sub(r'"(\\.|[^"\\])*"', '""', code)
        for a, b in ("{}", "()", "[]"):
            assert code.count(a) == code.count(b), (f, a)
    imported = set(re.findall(r"extern int (gymnet_[a-z_0-9]+)\(", srcs["Native.cs"])) | set(re.findall(r"extern IntPtr (gymnet_[a-z_0-9]+)\(", srcs["Native.cs"]))
    missing = declared - imported
    # device-side sampling helpers are not needed by the managed wrapper; everything else must be importable
    assert missing <= {"gymnet_sample_discrete_device", "gymnet_sample_box_device", "gymnet_vecenv_sample_actions_device",
                       "gymnet_vecenv_device_view"}, missing


def test_mirror_exposes_the_reference_member_names(gymnet):
    # IVecEnv / VecEnv (src/Gym/Envs/IVecEnv.cs:8-19, VecEnv.cs:12-93) and IEnv / Env (IEnv.cs:11-22, Env.cs:13-41)
    for m in ("Reset", "Step", "Close", "Seed", "StepAsync", "get_attr", "set_attr"):
        assert callable(getattr(gymnet.VectorEnv, m)), m
    assert issubclass(gymnet.DummyVecEnv, gymnet.VectorEnv)           # DummyVecEnv.cs:2-4
    for m in ("Reset", "Step", "StepAsync", "Render", "CloseEnvironment", "Seed", "Dispose"):
        assert callable(getattr(gymnet.CartPoleEnv, m)), m
    for m in ("Sample", "Contains", "Seed"):                       # Space.cs:15-17
        assert callable(getattr(gymnet.Box, m)) and callable(getattr(gymnet.Discrete, m)), m
    assert set(gymnet.Step.__slots__) == {"Observation", "Reward", "Done", "Information"}      # Step.cs:8-11

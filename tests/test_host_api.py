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
    assert lib.gymnet_abi_version() == 6 == gymnet._capi.ABI_VERSION


def abi_manifest():
    """sizeof / offsetof of every ABI struct field, printed by a C program compiled from the header itself
    (tools/abi_manifest.c) — the layout truth the bindings are checked against."""
    import json
    import subprocess
    import tempfile
    exe = os.path.join(tempfile.mkdtemp(prefix="gymnet_abi_"), "abi_manifest")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tools", "abi_manifest.c")], check=True)
    return json.loads(subprocess.run([exe], check=True, capture_output=True, text=True).stdout)


CTYPES_OF = {"gymnet_config": "Config", "gymnet_env_info": "EnvInfo", "gymnet_device_view": "DeviceView",
             "gymnet_counters": "Counters", "gymnet_rollout_buffers": "RolloutBuffers", "gymnet_group_config": "GroupConfig",
             "gymnet_ipc_handle": "IpcHandle", "gymnet_launch_policy": "LaunchPolicy", "gymnet_rollout_spec": "RolloutSpec"}


def test_struct_layouts_match_the_header(gymnet):
    man = abi_manifest()
    assert man["abi_version"] == gymnet._capi.ABI_VERSION
    for cname, pyname in CTYPES_OF.items():
        st = getattr(gymnet._capi, pyname)
        assert ctypes.sizeof(st) == man[cname]["size"], cname
        assert [f[0] for f in st._fields_] == [f[0] for f in man[cname]["fields"]], cname      # same fields, same order
        for fname, off, size in man[cname]["fields"]:
            d = getattr(st, fname)
            assert (d.offset, d.size) == (off, size), (cname, fname)
    # every struct the header defines is in the manifest (a new struct cannot dodge the check)
    hdr = open(os.path.join(ROOT, "include", "gymnet_amd.h")).read()
    assert set(re.findall(r"typedef struct (gymnet_\w+) \{", hdr)) == set(CTYPES_OF)


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
    # a state component the observation repeats verbatim is stored once, in the observation array (ABI 3)
    assert [gymnet.env_describe(k).traffic_bytes_per_step for k in (0, 1, 2, 3)] == [41, 33, 25, 57]
    assert list(gymnet.env_describe(3).state_row_in_obs) == [-1, -1, 4, 5, -1, -1, -1, -1]
    assert list(gymnet.env_describe(1).state_row_in_obs)[:2] == [-1, 2] and set(gymnet.env_describe(0).state_row_in_obs) == {-1}
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


def test_library_does_not_read_the_process_environment_for_its_launch_policy():
    """a host process's environment must not change which kernel the library runs.  Since round 5 there is no probe build
    either: `getenv` does not appear anywhere in the native sources, build.py has one flag set and one output, and the built
    library carries none of the old variable names."""
    csrc = os.path.join(ROOT, "gym.net_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, f)).read()
        assert "getenv" not in text and "GYMNET_PROBE_ENV" not in text, f
    build = open(os.path.join(ROOT, "gym.net_amd", "build.py")).read()
    assert "os.environ" not in build and "-DGYMNET" not in build
    import __graft_entry__ as ge
    blob = open(ge.load_package().LIB_PATH, "rb").read()
    assert b"GYMNET_RESET_FORM" not in blob and b"GYMNET_GRAPH" not in blob and b"GYMNET_VEC" not in blob


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


def test_box_sample_integer_dtypes_floor_like_the_reference(gymnet):
    """Box.cs:86-89: `if (DType == np.int32 || DType == np.uint32 || DType == np.@byte) sample = np.floor(sample)` and then
    astype(DType).  For those three dtypes negative samples round DOWN (uniform(-3, 0) never yields 0 except from -0.x -> -1);
    for any other integer dtype the reference only casts, which truncates toward zero."""
    i32 = gymnet.Box(-3.0, 0.0, (4000,), dtype=np.int32, seed=1).Sample()
    assert i32.dtype == np.int32 and set(np.unique(i32)) == {-3, -2, -1}                 # floor: (-1, 0) -> -1, 0 is never produced
    i64 = gymnet.Box(-3.0, 0.0, (4000,), dtype=np.int64, seed=1).Sample()
    assert i64.dtype == np.int64 and set(np.unique(i64)) == {-2, -1, 0}                  # cast only: (-1, 0) -> 0, (-3, -2) -> -2
    u8 = gymnet.Box(0.0, 3.0, (4000,), dtype=np.uint8, seed=2).Sample()
    assert u8.dtype == np.uint8 and set(np.unique(u8)) == {0, 1, 2}
    f = gymnet.Box(-3.0, 0.0, (1000,), dtype=np.float32, seed=3).Sample()
    assert f.dtype == np.float32 and (f != np.floor(f)).any()


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


CS_SIZES = {"uint": 4, "int": 4, "long": 8, "ulong": 8, "float": 4, "IntPtr": 8, "byte": 1}
CS_STRUCT_OF = {"gymnet_config": "GymnetConfig", "gymnet_env_info": "GymnetEnvInfo", "gymnet_device_view": "GymnetDeviceView",
                "gymnet_counters": "GymnetCounters", "gymnet_rollout_buffers": "GymnetRolloutBuffers",
                "gymnet_group_config": "GymnetGroupConfig", "gymnet_ipc_handle": "GymnetIpcHandle",
                "gymnet_launch_policy": "GymnetLaunchPolicy", "gymnet_rollout_spec": "GymnetRolloutSpec"}


def _csharp_sources():
    return {f: open(os.path.join(ROOT, "gym.net_amd", "csharp", f)).read() for f in ("Native.cs", "VectorEnv.cs", "GpuEnv.cs")}


def _cs_struct_layout(text, name):
    """[(field, offset, size)] of a [StructLayout(LayoutKind.Sequential)] struct under the default packing rules
    (every field at its natural alignment, struct size rounded up to the largest alignment)."""
    m = re.search(r"\[StructLayout\(LayoutKind\.Sequential\)\]\s*public (?:unsafe )?struct " + name + r"\s*\{([^{}]*)\}", text)
    assert m, name
    out, off, maxal = [], 0, 1
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        f = re.fullmatch(r"public (fixed )?(\w+) (\w+)(?:\[(\d+)\])?", decl)
        assert f, (name, decl)            # one field per declaration: `public int a, b;` would hide an order mistake
        size1 = CS_SIZES[f.group(2)]
        count = int(f.group(4)) if f.group(4) else 1
        off = (off + size1 - 1) // size1 * size1
        out.append((f.group(3), off, size1 * count))
        off += size1 * count
        maxal = max(maxal, size1)
    return out, (off + maxal - 1) // maxal * maxal


def test_csharp_binding_sources_lex_cleanly_and_cover_the_header():
    """No .NET toolchain exists here, so the C# binding cannot be compiled.  What can be checked: both files tokenize
    without a single error token (pygments' C# lexer), braces / parentheses balance, Native.cs declares a [DllImport] for
    EVERY function the header exports (no exceptions) and nothing the header does not, and every Sequential struct matches
    the C layout manifest field by field."""
    from pygments.lexers.dotnet import CSharpLexer
    from pygments.token import Error
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gymnet_amd.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(gymnet_[a-z_0-9]+)\s*\(", hdr))
    srcs = _csharp_sources()
    for f, text in srcs.items():
        toks = list(CSharpLexer().get_tokens(text))
        assert not [v for tt, v in toks if tt is Error], f
        code = re.sub(r"//.*", "", text)
        code = re.sub(r'"(\\.|[^"\\])*"', '""', code)
        for a, b in ("{}", "()", "[]"):
            assert code.count(a) == code.count(b), (f, a)
    imported = set(re.findall(r"\[DllImport\(Lib\)\] public static extern (?:int|IntPtr) (gymnet_[a-z_0-9]+)\(", srcs["Native.cs"]))
    assert declared - imported == set(), sorted(declared - imported)
    assert imported - declared == set(), sorted(imported - declared)
    # every native call the managed wrapper makes is one of the declared imports
    used = set(re.findall(r"Native\.(gymnet_[a-z_0-9]+)\(", srcs["VectorEnv.cs"] + srcs["GpuEnv.cs"]))
    assert used and used <= imported, sorted(used - imported)
    man = abi_manifest()
    for cname, csname in CS_STRUCT_OF.items():
        fields, size = _cs_struct_layout(srcs["Native.cs"], csname)
        assert size == man[cname]["size"], (csname, size, man[cname]["size"])
        assert fields == [tuple(f) for f in man[cname]["fields"]], csname
    assert set(CS_STRUCT_OF) == set(CTYPES_OF)
    # status / flag enums carry the header's values
    for cs, c in (("Rccl = -9", "GYMNET_ERR_RCCL = -9"), ("DoubleBuffer = 0x20", "GYMNET_FLAG_DOUBLE_BUFFER    0x20u"),
                  ("F64 = 0x40", "GYMNET_FLAG_F64              0x40u"), ("CompactRecordsOnly = 0x80", "GYMNET_FLAG_COMPACT_RECORDS_ONLY 0x80u"),
                  ("LaneSeeds = 8", "GYMNET_ARRAY_LANE_SEEDS = 8")):
        assert cs in srcs["Native.cs"] and c in open(os.path.join(ROOT, "include", "gymnet_amd.h")).read()


def _split_args(argtext):
    """Top-level comma split of an argument / parameter list."""
    out, depth, cur = [], 0, ""
    for ch in argtext:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return [a for a in out if a != "void"]


def _call_args(text, start):
    """The argument text of the call whose '(' is at text[start]."""
    depth = 0
    for i in range(start, len(text)):
        depth += text[i] == "("
        depth -= text[i] == ")"
        if depth == 0:
            return text[start + 1:i]
    raise AssertionError("unbalanced call")


def test_csharp_imports_and_call_sites_have_the_headers_arity():
    """Uncompiled C# again: every [DllImport] takes as many parameters as the C prototype it binds, and every
    Native.gymnet_*(...) call in VectorEnv.cs passes as many arguments as that import declares."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gymnet_amd.h")).read(), flags=re.S)
    c_arity = {}
    for m in re.finditer(r"\b(gymnet_[a-z_0-9]+)\s*\(", hdr):
        c_arity[m.group(1)] = len(_split_args(_call_args(hdr, m.end() - 1)))
    srcs = _csharp_sources()
    native = re.sub(r"//.*", "", srcs["Native.cs"])
    cs_arity = {}
    for m in re.finditer(r"public static extern (?:int|IntPtr) (gymnet_[a-z_0-9]+)\(", native):
        cs_arity[m.group(1)] = len(_split_args(_call_args(native, m.end() - 1)))
    assert cs_arity.keys() == c_arity.keys()
    wrong = {k: (cs_arity[k], c_arity[k]) for k in c_arity if cs_arity[k] != c_arity[k]}
    assert not wrong, wrong
    managed = re.sub(r"//.*", "", srcs["VectorEnv.cs"] + srcs["GpuEnv.cs"])
    calls = 0
    for m in re.finditer(r"Native\.(gymnet_[a-z_0-9]+)\(", managed):
        got = len(_split_args(_call_args(managed, m.end() - 1)))
        assert got == cs_arity[m.group(1)], (m.group(1), got, cs_arity[m.group(1)], managed[m.start():m.start() + 120])
        calls += 1
    assert calls >= 20


def test_ctypes_prototypes_have_the_headers_arity(gymnet):
    """The ctypes mirror declares argtypes for every exported function, with as many parameters as the C prototype."""
    from importlib import import_module
    capi = import_module(gymnet.__name__ + "._capi")
    lib = capi.load_library()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gymnet_amd.h")).read(), flags=re.S)
    wrong = {}
    for m in re.finditer(r"\b(gymnet_[a-z_0-9]+)\s*\(", hdr):
        want = len(_split_args(_call_args(hdr, m.end() - 1)))
        fn = getattr(lib, m.group(1))
        got = None if fn.argtypes is None else len(fn.argtypes)
        if got != want:
            wrong[m.group(1)] = (got, want)
    assert not wrong, wrong


# IVecEnv (src/Gym/Envs/IVecEnv.cs:8-19): the methods a polymorphic caller can reach; VecEnv implements Seed(int), Seed(int[])
# NON-virtually (VecEnv.cs:44-53) and StepAsync likewise (VecEnv.cs:63-65), Reset / Step / Close abstractly.
IVECENV_METHODS = {"Reset": "NDArray[] Reset()", "Step": "Step[] Step(int action)", "Close": "void Close()",
                   "Seed(int[])": "void Seed(int[] seed)", "Seed(int)": "void Seed(int seed)"}


def test_csharp_vectorenv_is_a_drop_in_through_the_reference_interface():
    """The round-1 binding hid VecEnv's non-virtual Seed / StepAsync with `public new` only: through an IVecEnv or VecEnv
    reference the BASE methods ran over an empty Environments list.  Checks on the source text:
    the class re-lists IVecEnv, implements both Seed overloads explicitly, overrides the abstract members, installs a
    non-empty Environments list whose lane proxies forward Seed, and routes StepAsync through the native async pair."""
    src = _csharp_sources()["VectorEnv.cs"]
    code = re.sub(r"//.*", "", src)
    head = re.search(r"public sealed unsafe class VectorEnv\s*:\s*([^{]+)\{", code)
    bases = [b.strip() for b in head.group(1).split(",")]
    assert bases[0] == "VecEnv" and "IVecEnv" in bases and "IDisposable" in bases
    assert re.search(r"void IVecEnv\.Seed\(int seed\)", code) and re.search(r"void IVecEnv\.Seed\(int\[\] seed\)", code)
    for m in ("public override NDArray[] Reset()", "public override Step[] Step(int action)", "public override void Close()"):
        assert m in code, m
    # a `new`-hidden member is only acceptable when the interface path is re-implemented explicitly (Seed) or the member
    # is not part of IVecEnv at all (StepAsync)
    for hidden in re.findall(r"public new [\w<>\[\]]+ (\w+)\(", code):
        assert hidden in ("Seed", "StepAsync"), hidden
    assert "Environments = new LaneList(this)" in code
    assert re.search(r"class LaneEnv\s*:\s*IEnv", code) and "public void Seed(int seed) => Owner.SeedLaneFromProxy(Lane, seed)" in code
    sa = code[code.index("public new Task<Step[]> StepAsync(int action)"):]
    sa = sa[:sa.index("void IVecEnv.Seed")]
    assert "gymnet_vecenv_step_async" in sa and "gymnet_vecenv_step_wait" in sa and "Step(action)" not in sa
    # the list above really is the reference's interface (checked where the reference tree exists: the build container)
    ref = "/root/reference/src/Gym/Envs/IVecEnv.cs"
    if os.path.exists(ref):
        text = open(ref, encoding="utf-8-sig").read()
        methods = set(re.findall(r"^\s+([\w\[\]]+ \w+\([^)]*\));", text, flags=re.M))
        assert methods == set(IVECENV_METHODS.values()), methods
        vec = open("/root/reference/src/Gym/Envs/VecEnv.cs", encoding="utf-8-sig").read()
        assert "public void Seed(int seed)" in vec and "public void Seed(int[] seed)" in vec            # non-virtual: cannot be overridden
        assert "public Task<Step[]> StepAsync(int action)" in vec and "public IList<IEnv> Environments { get; set; }" in vec


def test_csharp_single_instance_envs_derive_from_env_and_override_every_abstract_member():
    """SURVEY §8(f)-3 / VERDICT r2: `GpuCartPoleEnv : Env` (+ Pendulum / MountainCar / Acrobot) over a 1-lane handle, so that the
    reference's README loop (README.md:32-52) has a C# drop-in.  Source-text checks (no .NET here): the base derives from Env,
    overrides exactly the abstract members Env declares (Env.cs:20-30) plus the virtual StepAsync on the native async pair,
    returns a COPY from Reset like CartPoleEnv.cs:66, casts a Discrete action with `(int) action` (InvalidCastException like
    :138), and the optional TimeLimit reports truncation through Step.Information."""
    code = re.sub(r"//.*", "", _csharp_sources()["GpuEnv.cs"])
    assert re.search(r"public abstract unsafe class GpuEnv\s*:\s*Env\s*\{", code)
    for cls, env in (("GpuCartPoleEnv", "CartPole"), ("GpuPendulumEnv", "Pendulum"), ("GpuMountainCarEnv", "MountainCar"), ("GpuAcrobotEnv", "Acrobot")):
        assert re.search(r"public sealed class %s\s*:\s*GpuEnv\s*\{[^}]*base\(GymnetEnvId\.%s," % (cls, env), code), cls
    overrides = set(re.findall(r"public override ([\w<>\[\]]+ \w+)\(", code))
    assert overrides == {"NDArray Reset", "Step Step", "Task<Step> StepAsync", "Image Render", "void CloseEnvironment", "void Seed"}
    assert "(int) action" in code and ".Clone()" in code and "TimeLimit.truncated" in code and "max_episode_steps = maxEpisodeSteps" in code
    sa = code[code.index("public override Task<Step> StepAsync"):code.index("public override Image Render")]
    assert "gymnet_vecenv_step_async" in sa and "gymnet_vecenv_step_wait" in sa and "DistributedScheduler" not in sa
    ref = "/root/reference/src/Gym/Envs/Env.cs"
    if os.path.exists(ref):                                      # the abstract members really are these (build container only)
        text = open(ref, encoding="utf-8-sig").read()
        text = text[:text.index("public abstract class Env<TAction>")]
        abstract = set(re.findall(r"public abstract ([\w<>\[\]]+ \w+)\(", text))
        assert abstract == {"NDArray Reset", "Step Step", "Image Render", "void CloseEnvironment", "void Seed"}, abstract
        assert "public virtual Task<Step> StepAsync(object action)" in text


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

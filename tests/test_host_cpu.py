"""Host-side logic and the C-ABI surface, no GPU needed (no compute call is made)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pandas as pd
import pytest

import gym_trading_env_amd as gte
from gym_trading_env_amd import _abi, spaces, staging
from gym_trading_env_amd.config import make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "gte.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gte_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    names = _declared_functions()
    assert len(names) >= 18 and "gte_step" in names and "gte_reset" in names
    lib = _abi.load_library()
    assert sorted(_abi.SYMBOLS) == names, "ctypes table and include/gte.h disagree"
    for n in names:
        assert hasattr(lib, n), f"libgte.so does not export {n}"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _abi.LIB_PATH], text=True)
    exported = set(re.findall(r" T (gte_[a-z_0-9]+)", out))
    assert set(names) <= exported
    assert lib.gte_abi_version() == _abi.GTE_ABI_VERSION


def test_library_contains_gfx950_code_object_only():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={_abi.LIB_PATH}"], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets


def test_struct_layout_matches_the_c_header(oracle_mod):
    # the oracle is compiled against include/gte.h and rejects a config whose
    # struct_bytes differs from its sizeof(gte_config)
    cfg = make_config(n_envs=3, n_static=2)
    assert cfg.struct_bytes == C.sizeof(_abi.GteConfig)
    env = oracle_mod.OracleEnv(cfg, [(np.zeros((10, 4), np.float32), np.ones(10))])
    env.close()


@pytest.mark.skipif(_abi.load_library().gte_device_count() > 0, reason="host has a GPU")
def test_no_cpu_fallback_fails_loudly():
    """On a GPU-less host the product refuses to run instead of falling back."""
    from gym_trading_env_amd.batched import BatchedTradingEnv
    feat = np.zeros((50, 3), np.float32)
    with pytest.raises(gte.GteError) as ei:
        BatchedTradingEnv((feat, np.ones(50)), num_envs=4, output="numpy")
    assert ei.value.status == _abi.GTE_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(ImportError, match="no CPU fallback"):
        _abi.load_library(str(tmp_path / "libgte.so"))


def test_config_mirrors_reference_constructor_checks():
    # environments.py:106
    with pytest.raises(AssertionError, match="initial_position"):
        make_config(n_envs=1, n_static=1, positions=[0, 1], initial_position=0.5)
    cfg = make_config(n_envs=1, n_static=1, positions=[-1, 0, 1], initial_position=0,
                      portfolio_initial_value=5)
    assert cfg.initial_position_index == 1
    assert cfg.portfolio_initial_value == 5.0 and isinstance(cfg.portfolio_initial_value, float)  # :104
    assert make_config(n_envs=1, n_static=1).initial_position_index == -1  # 'random'
    assert make_config(n_envs=1, n_static=1).max_episode_duration == 0  # 'max'
    assert make_config(n_envs=1, n_static=1, windows=None).window == 0
    assert list(make_config(n_envs=1, n_static=1).positions[:2]) == [0.0, 1.0]  # default [0, 1] (:81)
    c = make_config(n_envs=1, n_static=1, reward_function=("clipped_log_return", 1.0, -0.002, 0.005))
    assert (c.reward_kind, c.reward_param1, c.reward_param2) == (_abi.REWARD_CLIPPED_LOG_RETURN, -0.002, 0.005)
    with pytest.raises(ValueError, match="unknown reward spec"):
        make_config(n_envs=1, n_static=1, reward_function="sharpe")
    with pytest.raises(ValueError, match="unknown dynamic feature"):
        make_config(n_envs=1, n_static=1, dynamic_feature_functions=["momentum"])


def test_callables_are_recognised_by_identity_never_by_name():
    """Only THIS package's default objects (and explicit string / tuple specs) map to the device
    enums.  A user's function that merely shares a default's NAME is the user's code: it must be
    evaluated (host / vectorised path), not silently replaced by the device built-in."""
    from gym_trading_env_amd import config, defaults

    def basic_reward_function(history):  # same name as the default, different semantics
        return 0.25

    def dynamic_feature_real_position(history):
        return 7.0

    def log_return(history):
        return -1.0

    assert config.resolve_reward(defaults.basic_reward_function)[0] == _abi.REWARD_LOG_RETURN
    assert config.resolve_reward("basic_reward_function")[0] == _abi.REWARD_LOG_RETURN
    for f in (basic_reward_function, log_return, lambda h: 0.0):
        assert config.resolve_reward(f)[0] == config.HOST_CALLABLE
    assert config.resolve_dynamic_features(
        [defaults.dynamic_feature_last_position_taken, defaults.dynamic_feature_real_position,
         "real_position", 0]) == [_abi.DYN_LAST_POSITION, _abi.DYN_REAL_POSITION,
                                  _abi.DYN_REAL_POSITION, _abi.DYN_LAST_POSITION]
    assert config.resolve_dynamic_features([dynamic_feature_real_position, lambda h: 0.0]) == \
        [config.HOST_CALLABLE, config.HOST_CALLABLE]
    # the public names of the package ARE the default objects
    import gym_trading_env_amd as g
    assert g.basic_reward_function is defaults.basic_reward_function
    assert g.dynamic_feature_real_position is defaults.dynamic_feature_real_position
    # make_config keeps a device placeholder for a user's callable (its value is overwritten
    # after every launch); the struct itself never names a host callable
    c = make_config(n_envs=1, n_static=1, reward_function=basic_reward_function,
                    dynamic_feature_functions=[dynamic_feature_real_position])
    assert c.reward_kind == _abi.REWARD_LOG_RETURN and c.n_dyn == 1
    with pytest.raises(TypeError):
        config.resolve_reward(3.5)


def _df(T=30):
    idx = pd.date_range("2021-01-01", periods=T, freq="h")
    rng = np.random.default_rng(0)
    return pd.DataFrame({"open": rng.random(T) + 1, "high": rng.random(T) + 2, "low": rng.random(T),
                         "close": rng.random(T) + 1, "volume": rng.random(T),
                         "feature_b": rng.random(T), "my_feature_a": rng.random(T)}, index=idx)


def test_stage_dataframe_follows_set_df():
    df = _df()
    s = staging.stage_dataframe(df, n_dyn=2)
    # :130 every column whose name CONTAINS "feature", DataFrame order
    assert s.feature_columns == ["feature_b", "my_feature_a", "dynamic_feature__0", "dynamic_feature__1"]
    assert s.feat.dtype == np.float32 and s.feat.shape == (30, 4) and s.feat.flags.c_contiguous
    np.testing.assert_array_equal(s.feat[:, 0], df["feature_b"].to_numpy().astype(np.float32))
    np.testing.assert_array_equal(s.feat[:, 2:], 0)                 # :135-138
    assert s.close.dtype == np.float64
    np.testing.assert_array_equal(s.close, df["close"].to_numpy())  # :143
    assert set(s.info_columns) == {"open", "high", "low", "close", "volume"}  # :131
    assert s.info_array.shape == (30, 5) and s.T == 30 and s.n_obs == 4
    with pytest.raises(KeyError):
        staging.stage_dataframe(df.drop(columns=["close"]))


def test_stage_arrays_and_geometry_checks():
    s = staging.stage_arrays(np.ones((20, 3)), np.arange(1, 21), n_dyn=1)
    assert s.feat.shape == (20, 4) and s.feat[:, 3].sum() == 0
    with pytest.raises(ValueError):
        staging.stage_arrays(np.ones((20, 3)), np.arange(19))
    staging.check_episode_geometry(100, 5, 50)
    with pytest.raises(ValueError, match="low >= high"):   # np.random.randint(low, high) (:174)
        staging.check_episode_geometry(60, 5, 55)
    with pytest.raises(ValueError, match="too short"):
        staging.check_episode_geometry(5, 5, "max")


def test_spaces():
    d = spaces.Discrete(3)
    assert d.n == 3 and d.contains(d.sample()) and not d.contains(3)
    b = spaces.Box(-np.inf, np.inf, shape=[4, 7])
    assert tuple(b.shape) == (4, 7) and b.dtype == np.float32
    md = spaces.MultiDiscrete([3] * 5)
    assert md.contains(md.sample())


def test_golden_fixtures_are_data_only():
    import replay
    names = replay.golden_names()
    assert len(names) >= 10
    for n in names:
        z = np.load(os.path.join(replay.GOLDEN_DIR, n + ".npz"), allow_pickle=False)
        assert "obs" in z.files and "cfg_json" in z.files
    z = np.load(os.path.join(replay.GOLDEN_DIR, "portfolio_random.npz"), allow_pickle=False)
    assert z["inputs"].shape == (3000, 9) and z["outputs"].shape == (3000, 6)


def _resource_usage(source):
    """VGPRs / scratch / occupancy per kernel from hipcc's resource-usage remarks (gfx950)."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "gym-trading-env_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                            "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(ROOT, "include"),
                            "-c", os.path.join(csrc, source), "-o", os.path.join(tmp, "x.o"),
                            "-Rpass-analysis=kernel-resource-usage"],
                           capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = []
    for block in r.stderr.split("Function Name:")[1:]:
        get = lambda key: int(re.search(key + r":\s*(\d+)", block).group(1))
        out.append({"name": block.split()[0], "vgprs": get("VGPRs"),
                    "scratch": get(r"ScratchSize \[bytes/lane\]"),
                    "occupancy": get(r"Occupancy \[waves/SIMD\]")})
    assert out, r.stderr[-2000:]
    return out


@pytest.mark.parametrize("source,min_occupancy", [("gte_hot.hip", 6), ("gte_hot_nt.hip", 6),
                                                  ("gte_rollout.hip", 4)])
def test_kernel_register_budget(source, min_occupancy):
    """The step kernel's speed hangs on its register allocation: one occupancy step costs
    10+ us per step, and scratch use doubles the store traffic.  A change to the
    shared device code that pushes the hot instantiations over their budget fails here, on the
    CPU, before anything is measured."""
    for k in _resource_usage(source):
        assert k["scratch"] == 0, k
        assert k["occupancy"] >= min_occupancy, k


def test_features_compiled_out_of_the_hot_translation_units_are_guarded():
    """The isolated hot instantiations (GTE_HOT_ONLY) leave features out of phase A.  Every such
    block must name the Params field it depends on, that field must be tested by
    `hot_tu_covers` (gte_device.h), `gte_step` must pick the kernel with that predicate, and every
    launcher of a GTE_HOT_ONLY translation unit must refuse a launch it does not cover — so a
    launch predicate cannot drift from a compiled-out feature again (round 2: `final_info` read
    records the kernel never wrote)."""
    csrc = os.path.join(ROOT, "gym-trading-env_amd", "csrc")
    read = lambda f: open(os.path.join(csrc, f)).read()
    dev = read("gte_device.h")
    body = re.search(r"inline bool hot_tu_covers\(const Params& p\) \{(.*?)\}", dev, re.S).group(1)
    kernels = read("gte_kernels.hip")
    # blocks inside device code carry `// p.<field>:`; the two file-scope blocks (host launchers,
    # helper kernels of the shared TU) do not touch phase A
    blocks = re.findall(r"#ifndef GTE_HOT_ONLY(.*)", kernels)
    fields = [re.match(r"\s*//\s*p\.([a-z_]+)", b) for b in blocks]
    named = [m.group(1) for m in fields if m]
    assert len(blocks) - len(named) == 2, "an #ifndef GTE_HOT_ONLY block inside phase A does not name its field"
    assert sorted(set(named)) == ["final_rec", "log"]
    for f in set(named):
        assert re.search(rf"p\.{f}\b", body), f"hot_tu_covers does not test p.{f}"
    api = read("gte_api.hip")
    assert re.search(r"const bool hot = [^;]*gte::hot_tu_covers\(p\)", api)
    for tu in ("gte_hot.hip", "gte_rollout.hip"):
        src = read(tu)
        assert "#define GTE_HOT_ONLY 1" in src
        launchers = re.findall(r"hipError_t (?:GTE_HOT_NAME\()?(launch_[a-z_]+)\)?\(const Params& p.*?\n\}", src, re.S)
        assert launchers
        for m in re.finditer(r"hipError_t (?:GTE_HOT_NAME\()?launch_[a-z_]+\)?\(const Params& p.*?\n\}", src, re.S):
            assert "if (!hot_tu_covers(p)) return hipErrorInvalidValue;" in m.group(0), m.group(0)[:80]


def test_inline_asm_wide_stores_carry_their_wait_states():
    """A VMEM store of more than 64 bits keeps reading its data VGPRs for a couple of cycles after
    it issues; hipcc pads its own stores but does not look inside inline asm, so an asm
    `global_store_dwordx3/x4` must end with `s_nop 1` INSIDE its string — otherwise the compiler's
    next VALU instruction may overwrite the data before the store has read it (round 3: the lean
    copy loop stored address words in place of x, y until this was added; the generic loop had
    been getting away with it by luck of scheduling)."""
    csrc = os.path.join(ROOT, "gym-trading-env_amd", "csrc")
    found = 0
    for name in os.listdir(csrc):
        if not name.endswith((".hip", ".h")):
            continue
        src = open(os.path.join(csrc, name)).read()
        for m in re.finditer(r'asm\s+volatile\(\s*"([^"]*(?:"\s*"[^"]*)*)"', src):
            text = m.group(1)
            if re.search(r"(global|buffer|flat)_store_dwordx[34]", text):
                found += 1
                assert re.search(r"_store_dwordx[34][^\\]*\\n\\ts_nop 1", text), (name, text)
    assert found >= 1


def test_integration_md_stub_matches_the_abi():
    """The ctypes stub INTEGRATION.md shows a maintainer of the reference is not prose: its
    gte_config must have the fields, order and size of the real one."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = text.split("```python")[1].split("```")[0]
    code = code.split('lib = C.CDLL("libgte.so")')[0]  # the struct, not the load
    scope = {}
    exec(compile(code, "INTEGRATION.md", "exec"), scope)  # our own document
    stub = scope["gte_config"]
    assert [f[0] for f in stub._fields_] == [f[0] for f in _abi.GteConfig._fields_]
    assert C.sizeof(stub) == C.sizeof(_abi.GteConfig)
    for (name, a), (_, b) in zip(stub._fields_, _abi.GteConfig._fields_):
        assert C.sizeof(a) == C.sizeof(b), name


def test_stage_dataframe_equals_the_reference_set_df():
    """tests/golden/set_df.npz: what the reference's `_set_df` made of an awkward DataFrame
    (feature columns = name contains 'feature', case-sensitive; the rest is info)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "set_df.npz"), allow_pickle=False)
    cols = [str(c) for c in z["columns"]]
    df = pd.DataFrame(z["values"], columns=cols,
                      index=pd.date_range("2021-03-01", periods=len(z["values"]), freq="h"))
    s = staging.stage_dataframe(df, n_dyn=2)
    assert s.feature_columns == [str(c) for c in z["features_columns"]]
    assert s.n_obs == int(z["nb_features"]) and s.n_static == int(z["nb_static_features"])
    np.testing.assert_array_equal(s.feat, z["obs_array"])
    np.testing.assert_array_equal(s.close, z["price_array"])
    ref_info = [str(c) for c in z["info_columns"]]  # a set difference: order is arbitrary there
    assert sorted(s.info_columns) == sorted(ref_info)
    for j, c in enumerate(ref_info):
        np.testing.assert_array_equal(np.asarray(s.info_array[:, s.info_columns.index(c)], np.float64),
                                      z["info_array"][:, j])


@pytest.mark.parametrize("modern", [True, False])
def test_gymnasium_registration_with_a_stand_in(monkeypatch, modern):
    """register_gymnasium_ids() (the reference's __init__.py:3-14: same ids, same flags) against
    an in-memory stand-in of `gymnasium.envs.registration` — gymnasium itself is not installed
    here.  Gymnasium >= 1.0 also gets a vector entry point (gym.make_vec -> one batched env);
    an older register() without that keyword is called the reference's way."""
    import sys
    import types
    import gym_trading_env_amd as gte
    calls = []

    def register_modern(id, entry_point=None, vector_entry_point=None, disable_env_checker=False,
                        order_enforce=True, **kw):
        calls.append(dict(id=id, entry_point=entry_point, vector_entry_point=vector_entry_point,
                          disable_env_checker=disable_env_checker, order_enforce=order_enforce))

    def register_old(id, entry_point=None, disable_env_checker=False, order_enforce=True):
        calls.append(dict(id=id, entry_point=entry_point, vector_entry_point=None,
                          disable_env_checker=disable_env_checker, order_enforce=order_enforce))

    reg = types.ModuleType("gymnasium.envs.registration")
    reg.register = register_modern if modern else register_old
    reg.registry = {}
    gym = types.ModuleType("gymnasium")
    envs_mod = types.ModuleType("gymnasium.envs")
    envs_mod.registration = reg
    gym.envs = envs_mod
    for name, mod in (("gymnasium", gym), ("gymnasium.envs", envs_mod),
                      ("gymnasium.envs.registration", reg)):
        monkeypatch.setitem(sys.modules, name, mod)
    assert gte.register_gymnasium_ids() is True
    assert [c["id"] for c in calls] == ["TradingEnv", "MultiDatasetTradingEnv"]
    from gym_trading_env_amd import envs
    assert calls[0]["entry_point"] is envs.TradingEnv and calls[1]["entry_point"] is envs.MultiDatasetTradingEnv
    assert all(c["disable_env_checker"] is True and c["order_enforce"] is False for c in calls)
    assert all((c["vector_entry_point"] is not None) == modern for c in calls)


def _build_c_demo(tmp_path):
    import subprocess
    exe = str(tmp_path / "c_abi_demo")
    csrc = os.path.join(ROOT, "gym-trading-env_amd", "csrc")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", csrc, "-lgte",
                           f"-Wl,-rpath,{csrc}", "-lm", "-o", exe])
    return exe


def test_c_program_binds_the_abi_and_fails_loudly_without_a_gpu(tmp_path):
    """examples/c_abi_demo.c: a plain C host (gcc, include/gte.h, -lgte; no Python, no torch)
    compiles against the boundary; on a host without a gfx950 device it stops at gte_create with
    GTE_ERR_NO_DEVICE — there is no CPU fallback to fall into."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the C demo runs for real in tests/test_gpu_examples.py")
    exe = _build_c_demo(tmp_path)
    r = subprocess.run([exe, "64", "5"], capture_output=True, text=True)
    assert r.returncode == 3
    assert "no CPU fallback" in r.stderr and "gte_create" in r.stderr

"""Host-side logic that needs no GPU: registry, kwargs, spaces, sharding arithmetic, and the
rule that the product never reaches into oracle/ or falls back to a CPU path."""
import os
import re

import numpy as np
import pytest

import gym_fishing_amd as gf
from conftest import ROOT
from gym_fishing_amd import sharding, spaces


def test_registry_matches_reference_ids():
    # gym_fishing/envs/__init__.py:17-71 (there is no fishing-v3)
    assert gf.ENV_IDS == ("fishing-v0", "fishing-v1", "fishing-v2", "fishing-v4", "fishing-v5", "fishing-v6",
                          "fishing-v7", "fishing-v8", "fishing-v9", "fishing-v10", "fishing-v11")
    assert [gf.env_class("fishing-v%d" % k).__name__ for k in range(5, 12)] == [
        "Allen", "BevertonHolt", "May", "Myers", "Ricker", "NonStationary", "ModelUncertainty"]
    assert gf.env_class("fishing-v0").__name__ == "FishingEnv"
    assert gf.env_class("fishing-v1").__name__ == "FishingCtsEnv"
    assert gf.env_class("fishing-v2").__name__ == "FishingTippingEnv"
    assert gf.env_class("fishing-v4").__name__ == "FishingModelError"
    with pytest.raises(KeyError):
        gf.env_class("fishing-v3")


def test_constructor_kwargs_match_reference_signatures():
    import inspect
    want = {
        "fishing-v0": dict(r=0.3, K=1, sigma=0.0, n_actions=100, init_state=0.75, Tmax=100, file=None),
        "fishing-v1": dict(r=0.3, K=1, sigma=0.0, init_state=0.75, Tmax=100, file=None),
        "fishing-v2": dict(r=0.3, K=1, C=0.5, sigma=0.0, init_state=0.75, Tmax=100, file=None),
        "fishing-v4": dict(K_mean=1.0, r_mean=0.3, price=1.0, sigma=0.0, sigma_p=0.1, init_state=0.75, Tmax=100,
                           file=None),
        # growth_models.py:6-154
        "fishing-v5": dict(r=0.3, K=1, C=0.5, sigma=0.0, init_state=0.75, Tmax=100, file=None),
        "fishing-v6": dict(r=0.3, K=1, sigma=0.0, init_state=0.75, Tmax=100, file=None),
        "fishing-v7": dict(r=0.7, K=1.5, M=1.5, q=3, b=0.15, sigma=0.0, a=0.2, init_state=0.75, Tmax=100, file=None),
        "fishing-v8": dict(r=1.0, K=1.0, M=1.0, theta=3.0, sigma=0.0, init_state=1.5, Tmax=100, file=None),
        "fishing-v9": dict(r=0.3, K=1, sigma=0.0, init_state=0.75, Tmax=100, file=None),
        "fishing-v10": dict(r=0.8, K=1, sigma=0.0, alpha=-0.007, init_state=0.75, Tmax=100, file=None),
    }
    for env_id, kw in want.items():
        sig = inspect.signature(gf.env_class(env_id).__init__)
        got = {k: v.default for k, v in sig.parameters.items() if v.kind == v.POSITIONAL_OR_KEYWORD and k != "self"}
        assert got == kw, env_id
        assert list(got) == list(kw), "positional order differs for " + env_id


def test_env_classes_carry_the_reference_method_names():
    # base_fishing_env.py:60-164: the public methods of BaseFishingEnv, with the reference's argument names
    import inspect
    want = {"step": ["action"], "reset": [], "render": ["mode"], "close": [], "simulate": ["model", "reps"],
            "plot": ["df", "output"], "policyfn": ["model", "reps"], "plot_policy": ["df", "output"],
            "harvest_draw": ["quota"], "population_draw": [], "get_quota": ["action"], "get_action": ["quota"],
            "get_fish_population": ["state"], "get_state": ["fish_population"]}
    for env_id in gf.ENV_IDS:
        cls = gf.env_class(env_id)
        for name, args in want.items():
            fn = getattr(cls, name, None)
            assert callable(fn), (env_id, name)
            params = [k for k in inspect.signature(fn).parameters if k != "self"]
            assert params[:len(args)] == args, (env_id, name, params)
            # whatever this build adds behind the reference's arguments is optional
            extra = list(inspect.signature(fn).parameters.values())[1 + len(args):]
            assert all(q.default is not q.empty or q.kind in (q.VAR_KEYWORD, q.VAR_POSITIONAL, q.KEYWORD_ONLY) for q in extra), (env_id, name)


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(gf.FishingLibraryError, match="no CPU backend"):
        gf.make("fishing-v1", num_envs=8)
    with pytest.raises(gf.FishingLibraryError):
        gf.make("fishing-v0")


def test_product_never_imports_the_oracle_or_numpy_math_fallbacks():
    pkg = os.path.join(ROOT, "gym_fishing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".hip", ".h")):
                continue
            src = open(os.path.join(dirpath, f)).read()
            code = "\n".join(line.split("#")[0] for line in src.splitlines())
            assert not re.search(r"^\s*(from|import)\s+oracle\b", code, flags=re.M), f
            assert "scalar_env" not in code and "c_oracle" not in code, f


def test_spaces():
    b = spaces.Box(np.array([-1], dtype=np.float32), np.array([1], dtype=np.float32), dtype=np.float32)
    assert b.shape == (1,) and b.dtype == np.float32 and b.low[0] == -1 and b.high[0] == 1
    b.seed(0)
    s = b.sample()
    assert s.shape == (1,) and s.dtype == np.float32 and b.contains(s)
    assert not b.contains(np.array([1.5], dtype=np.float32))
    d = spaces.Discrete(100)
    d.seed(0)
    assert d.n == 100 and d.contains(d.sample()) and not d.contains(100) and not d.contains(-1)
    assert spaces.is_discrete(d) and not spaces.is_discrete(b)
    Box, Discrete = spaces.space_classes()
    assert Box is not None and Discrete is not None


@pytest.mark.parametrize("total,world", [(1 << 22, 1), (1 << 22, 8), (1 << 24, 8), (1000003, 8), (10, 4), (3, 2),
                                         (0, 3), (4, 8)])
def test_shard_range_tiles_the_batch(total, world):
    covered = 0
    prev_end = 0
    for rank in range(world):
        off, cnt = sharding.shard_range(total, rank, world)
        assert off % 4 == 0 or cnt == 0
        assert off == prev_end or cnt == 0
        assert cnt >= 0
        prev_end = off + cnt if cnt else prev_end
        covered += cnt
    assert covered == total
    counts = [sharding.shard_range(total, r, world)[1] for r in range(world)]
    assert max(counts) - min(counts) <= 7      # one 4-env unit + the trimmed tail of the last shard


def test_shard_range_rejects_bad_ranks():
    with pytest.raises(ValueError):
        sharding.shard_range(16, 2, 2)
    with pytest.raises(ValueError):
        sharding.shard_range(-1, 0, 1)


def test_summarize_record():
    import torch
    rec = torch.tensor([10.0, 30.0, 4.0, 400.0], dtype=torch.float64)
    s = sharding.summarize_record(rec)
    assert s["mean_return"] == 2.5 and s["mean_length"] == 100.0
    assert abs(s["std_return"] - np.sqrt(30 / 4 - 2.5 ** 2)) < 1e-12
    assert "mean_return" not in sharding.summarize_record(torch.zeros(4, dtype=torch.float64))


def test_register_with_gym_is_optional():
    assert isinstance(gf.register_with_gym(), list)     # [] when neither gym nor gymnasium exists


def test_each_registry_gets_the_api_it_speaks(monkeypatch):
    """Stand-in `gym` and `gymnasium` registries (neither package is installed here): the old gym is handed the
    reference's 4-tuple classes, as gym_fishing/envs/__init__.py:17-35 registers them; gymnasium -- whose checker and
    wrappers reject a 4-tuple step() -- the 5-tuple GymnasiumFishingEnv with the id as its constructor argument."""
    import sys
    import types
    seen = {"gym": [], "gymnasium": []}
    for mod in seen:
        top, envs, reg = types.ModuleType(mod), types.ModuleType(mod + ".envs"), types.ModuleType(mod + ".envs.registration")
        reg.register = (lambda mod: lambda id, entry_point, kwargs=None, **kw: seen[mod].append((id, entry_point, kwargs)))(mod)
        top.envs, envs.registration = envs, reg
        for name, m in ((mod, top), (mod + ".envs", envs), (mod + ".envs.registration", reg)):
            monkeypatch.setitem(sys.modules, name, m)
    done = gf.register_with_gym()
    assert sorted(done) == sorted((m, i) for m in seen for i in gf.ENV_IDS)
    assert ("fishing-v1", "gym_fishing_amd.envs:FishingCtsEnv", None) in seen["gym"]
    assert ("fishing-v4", "gym_fishing_amd.envs:FishingModelError", None) in seen["gym"]
    assert all(ep == "gym_fishing_amd.gymnasium_api:GymnasiumFishingEnv" and kw == {"env_id": i} for i, ep, kw in seen["gymnasium"])
    assert sorted(i for i, _, _ in seen["gymnasium"]) == sorted(gf.ENV_IDS)
    # the entry point gymnasium.make() would import and call exists, and takes the id + the reference's kwargs
    import importlib
    import inspect
    cls = getattr(importlib.import_module("gym_fishing_amd.gymnasium_api"), "GymnasiumFishingEnv")
    assert {"env_id", "kwargs"} <= set(inspect.signature(cls.__init__).parameters)
    assert inspect.signature(cls.reset).parameters.keys() >= {"seed", "options"}
    with pytest.raises(ValueError):
        gf.make("fishing-v1", api="gym5")
    with pytest.raises(KeyError):
        gf.make("fishing-v3", api="gymnasium")


def test_only_tests_smoke_and_bench_touch_the_oracle():
    """The oracle is the checker, never the thing shipped: outside tests/ and oracle/ itself, only
    bench.py (cpu_baseline leg) and __graft_entry__.py (build of the C restatement, smoke check)
    may import it."""
    allowed = {os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")}
    offenders = []
    for dirpath, dirs, files in os.walk(ROOT):
        # (dot-directories are tool state and scratch -- .git, caches, a git worktree of an earlier round kept for A/B
        # timing -- not part of the product)
        dirs[:] = [d for d in dirs if d not in ("tests", "oracle", "gpurun_out", "__pycache__") and not d.startswith(".")]
        for f in files:
            if not f.endswith(".py"):
                continue
            path = os.path.join(dirpath, f)
            code = "\n".join(line.split("#")[0] for line in open(path).read().splitlines())
            if re.search(r"^\s*(from|import)\s+oracle\b", code, flags=re.M) and path not in allowed:
                offenders.append(os.path.relpath(path, ROOT))
    assert not offenders, offenders


def test_build_keeps_every_translation_unit_its_own_device_link_and_csrc_free_of_build_variants():
    """(1) fishing_common.h's inline device templates have different bodies per translation unit (FISHING_ZOO_F64_FAR: 1 in
    fishing_aux.hip, 0 in fishing_step.hip / fishing_rollout.hip) -- sound only while no device symbol crosses units, i.e. while the
    library is built without relocatable device code; and -ffp-contract=off is part of the numerical contract.  (2) The product
    kernels carry no experiment scaffolding: a handful of preprocessor conditionals (the partial-slot count, three byte thresholds
    of the launch's walk, that per-unit setting), nothing else."""
    import glob
    import re
    from gym_fishing_amd import build
    assert "-fno-gpu-rdc" in build.HIPCC_FLAGS and "-ffp-contract=off" in build.HIPCC_FLAGS
    src = {f: open(f).read() for f in glob.glob(os.path.join(build.CSRC, "*"))}
    units = {os.path.basename(f): re.findall(r"^#define FISHING_ZOO_F64_FAR (\d)", s, re.M) for f, s in src.items() if f.endswith(".hip")}
    assert units == {"fishing_step.hip": ["0"], "fishing_rollout.hip": ["0"], "fishing_aux.hip": []}, units
    conditionals = [ln.strip() for s in src.values() for ln in s.splitlines() if re.match(r"\s*#\s*(if|ifdef|ifndef|elif)\b", ln)]
    assert len(conditionals) <= 15, conditionals
    knobs = sorted(set(re.findall(r"#\s*ifn?def\s+(FISHING_\w+)", "\n".join(src.values()))))
    assert knobs == ["FISHING_F64_E2_MAX_BYTES", "FISHING_NTA_MIN_BYTES", "FISHING_PARTIAL_SLOTS", "FISHING_XZZ_MIN_BYTES",
                     "FISHING_ZOO_F64_FAR"], knobs
    assert len(src[os.path.join(build.CSRC, "fishing_common.h")].splitlines()) < 1400

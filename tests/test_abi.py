"""CPU: the C-ABI library loads and exports every symbol include/pogema_amd.h declares; argument
validation and error reporting work without a GPU (no compute call is made)."""
import ctypes as C
import os
import re

import pytest

from pogema_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pogema_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgx_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(engine_lib):
    names = declared_symbols()
    assert len(names) >= 12
    for name in names:
        assert hasattr(engine_lib, name), f"{name} declared in include/pogema_amd.h but not exported"
    assert set(names) == set(_lib.EXPORTED_SYMBOLS)


def test_abi_version(engine_lib):
    assert engine_lib.pgx_abi_version() == _lib.PGX_ABI_VERSION


def test_config_struct_layout():
    # must match `struct pgx_config` in the header: 10 x int32, uint64, int64, int32, float, 6 x int32 (the last one: abi_version, ABI 6)
    assert C.sizeof(_lib.PgxConfig) == 10 * 4 + 8 + 8 + 4 + 4 + 6 * 4
    assert _lib.PgxConfig.soft_occupancy.offset == 80 and _lib.PgxConfig.abi_version.offset == 84
    assert _lib.PgxConfig.soft_vertex_rule.offset == 64 and _lib.PgxConfig.bad_action.offset == 72
    assert _lib.PgxConfig.lifelong_rng.offset == 76
    assert _lib.PgxConfig.seed.offset == 40 and _lib.PgxConfig.env_index_base.offset == 48
    assert _lib.PgxConfig.random_outside.offset == 56 and _lib.PgxConfig.outside_density.offset == 60


@pytest.mark.parametrize("field,value,needle", [
    ("batch", 0, "batch"), ("num_agents", 0, "num_agents"), ("num_agents", 5000, "num_agents"),
    ("obs_radius", 0, "obs_radius"), ("obs_radius", 16, "obs_radius"), ("collision_system", 7, "collision"),
    ("on_target", -1, "on_target"), ("height", 0, "map size"), ("width", 4096, "map size"),
    ("soft_vertex_rule", 2, "semantics"), ("coop_reward", -1, "semantics"), ("bad_action", 3, "semantics"),
    ("lifelong_rng", 2, "semantics"), ("soft_occupancy", 2, "semantics"), ("abi_version", 0, "abi_version"),
    ("abi_version", 5, "renumbered soft_occupancy"), ("abi_version", 7, "abi_version"),
    ("obs_dtype", 4, "obs_dtype"), ("obs_dtype", -1, "obs_dtype"),
])
def test_create_rejects_bad_config(engine_lib, field, value, needle):
    cfg = _lib.PgxConfig(batch=4, height=8, width=8, num_agents=2, obs_radius=3, collision_system=0, on_target=0,
                         max_episode_steps=64, auto_reset=0, obs_dtype=0, seed=0, env_index_base=0,
                         abi_version=_lib.PGX_ABI_VERSION)
    setattr(cfg, field, value)
    handle = C.c_void_p()
    status = engine_lib.pgx_create(C.byref(cfg), 0, C.byref(handle))
    assert status == -1 and not handle.value
    assert needle in engine_lib.pgx_last_error().decode()


def test_lds_limit_is_reported(engine_lib):
    cfg = _lib.PgxConfig(batch=1, height=1024, width=1024, num_agents=64, obs_radius=15, collision_system=0,
                         on_target=0, max_episode_steps=64, auto_reset=0, obs_dtype=0, seed=0, env_index_base=0,
                         abi_version=_lib.PGX_ABI_VERSION)
    handle = C.c_void_p()
    assert engine_lib.pgx_create(C.byref(cfg), 0, C.byref(handle)) == -1
    assert "LDS" in engine_lib.pgx_last_error().decode()


def test_buffers_info_struct_layout():
    # struct pgx_buffers_info: int64, 4 x int32, 2 x float, double, 2 x float
    assert C.sizeof(_lib.PgxBuffersInfo) == 8 + 16 + 8 + 8 + 8
    assert C.sizeof(_lib.PgxRolloutIO) == 8 * 8 + 4 + 4 + 8 + 8 + 8 + 8  # pgx_rollout_io
    assert _lib.PgxBuffersInfo.same_zone_us.offset == 24 and _lib.PgxBuffersInfo.spacer_gib.offset == 32
    assert _lib.PgxBuffersInfo.buffer_gbs.offset == 40


def test_null_arguments(engine_lib):
    assert engine_lib.pgx_step(None, None, 0, None, None, None, None, None, None) == -1
    assert engine_lib.pgx_observe(None, None, None) == -1
    assert engine_lib.pgx_destroy(None) == 0


def test_no_device_fails_loudly_not_silently(engine_lib):
    """Without a GPU the product must raise, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pogema_amd import GridConfig, VecPogema
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        VecPogema(GridConfig(num_agents=2), batch=2)
    cfg = _lib.PgxConfig(batch=4, height=8, width=8, num_agents=2, obs_radius=3, collision_system=0, on_target=0,
                         max_episode_steps=64, auto_reset=0, obs_dtype=0, seed=0, env_index_base=0,
                         abi_version=_lib.PGX_ABI_VERSION)
    handle = C.c_void_p()
    assert engine_lib.pgx_create(C.byref(cfg), 0, C.byref(handle)) == -2  # PGX_E_HIP
    assert not handle.value


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under pogema_amd/ may import, load or link it."""
    pkg = os.path.join(ROOT, "pogema_amd")
    banned = ("import oracle", "from oracle", "libpogema_oracle", "pogema_oracle", "c_oracle")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for word in banned:
                    assert word not in text, f"{f} references {word!r}"


def _cfg(**kw):
    base = dict(batch=4, height=16, width=16, num_agents=4, obs_radius=5, collision_system=0, on_target=0, max_episode_steps=64,
                auto_reset=1, obs_dtype=0, seed=0, env_index_base=0, random_outside=0, outside_density=0.3, soft_vertex_rule=0,
                coop_reward=0, bad_action=0, lifelong_rng=0, soft_occupancy=0, abi_version=_lib.PGX_ABI_VERSION)
    base.update(kw)
    return _lib.PgxConfig(**base)


def test_check_config_knows_the_limits_without_a_device(engine_lib):
    """pgx_check_config = every check of pgx_create that needs no GPU (ranges, PGX_MAX_*, the LDS budget of the launch shape,
    large-map layout included), same status codes and messages -- README.md "Limits" (VERDICT r5 missing #4)."""
    import ctypes as C

    def chk(**kw):
        cfg = _cfg(**kw)
        rc = engine_lib.pgx_check_config(C.byref(cfg))
        return rc, engine_lib.pgx_last_error().decode() if rc else ""

    assert chk() == (0, "")
    assert chk(height=1024, width=1024, num_agents=256)[0] == 0, "PGX_MAX_SIDE fits in the large-map layout"
    assert chk(height=800, width=1000, num_agents=70)[0] == 0
    assert chk(height=1024, width=1024, num_agents=1024, obs_radius=7)[0] == 0
    rc, msg = chk(height=1024, width=1024, num_agents=1024, obs_radius=15)
    assert rc != 0 and "bytes of LDS" in msg and "Limits" in msg
    assert chk(num_agents=1024, obs_radius=7, height=64, width=64)[0] == 0
    assert chk(num_agents=1024, obs_radius=8, height=64, width=64)[0] != 0, "1024 agents x 32-bit row masks exceed one CU"
    for bad, frag in ((dict(num_agents=1025), "num_agents"), (dict(obs_radius=16), "obs_radius"), (dict(height=1025), "map size"),
                      (dict(abi_version=5), "abi_version")):
        rc, msg = chk(**bad)
        assert rc != 0 and frag in msg, (bad, msg)


def test_vecpogema_names_the_limit_before_it_touches_a_device():
    """What GridConfig admits and the engine does not is a ValueError from VecPogema.__init__ that says the limit -- raised
    before the HIP device is even looked for (this container has none)."""
    from pogema_amd import GridConfig, VecPogema
    with pytest.raises(ValueError, match=r"obs_radius=20.*1\.\.15"):
        VecPogema(GridConfig(size=16, num_agents=2, obs_radius=20), batch=2)
    with pytest.raises(ValueError, match=r"num_agents=1500.*1\.\.1024"):
        VecPogema(GridConfig(size=64, num_agents=1500, obs_radius=2), batch=2)

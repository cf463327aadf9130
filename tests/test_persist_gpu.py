"""GPU: step_persist_kernel (persistent 3-wave workgroups: wave 0 produces slice i while waves 1-2 stream slice i-1) is
bit-identical with the C oracle -- all collision systems and episode modes, ragged shares (every workgroup takes 2-4
slices, some none), misaligned slices (odd agent counts), semantics variants, observe(), and in a long soak."""
import numpy as np
import pytest

from util import assert_rollouts_equal, c_oracle_rollout, engine_rollout, generate_instances, random_actions

pytestmark = pytest.mark.gpu


@pytest.fixture()
def forced_persist(monkeypatch):
    # PGX_PERSIST=2: persistent workgroups whatever the launch size; one workgroup per CU -> 256 workgroups for B slices
    monkeypatch.setenv("PGX_PERSIST", "2")
    monkeypatch.setenv("PGX_PERSIST_PER_CU", "1")


CASES = [  # name, B, H, W, A, r, density, T, max_steps
    ("full_wave", 700, 20, 20, 64, 5, 0.2, 12, 8),
    ("odd_agents", 531, 18, 22, 33, 4, 0.25, 10, 6),   # slice not a multiple of 16 bytes: head / tail floats
    ("wide_window", 300, 24, 24, 40, 7, 0.2, 8, 5),
    ("few_slices", 260, 16, 16, 48, 3, 0.15, 8, 64),   # most workgroups take one slice, four take two
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("collision", ["priority", "block_both", "soft"])
@pytest.mark.parametrize("on_target", ["finish", "restart", "nothing"])
def test_persistent_kernel_equals_c_oracle(forced_persist, case, collision, on_target):
    name, B, H, Wd, A, r, density, T, max_steps = case
    obstacles, agents, targets = generate_instances(B, H, Wd, A, density, seed=hash(name) % 1000 + 11)
    actions = random_actions(T, B, A, seed=5)
    kw = dict(obs_radius=r, collision_system=collision, on_target=on_target, max_episode_steps=max_steps, auto_reset=True,
              seed=9, env_index_base=3)
    ref = c_oracle_rollout(obstacles, agents, targets, actions, nthreads=8, **kw)
    got = engine_rollout(obstacles, agents, targets, actions, action_dtype="int8", **kw)
    assert_rollouts_equal(ref, got, f"persist/{name}/{collision}/{on_target}")


def test_persistent_kernel_is_what_ran_and_matches_the_regular_kernel(forced_persist, monkeypatch):
    """Same states through both launch shapes: identical tensors; pgx_debug_launch_shape says which kernel each handle
    launches (persistent: 256 workgroups of 3 waves for 900 slices)."""
    import torch
    from pogema_amd import GridConfig, VecPogema
    gc = GridConfig(size=24, num_agents=64, obs_radius=5, density=0.25, seed=4, collision_system="soft", max_episode_steps=16)
    B = 900
    a = VecPogema(gc, batch=B, auto_reset=True)
    monkeypatch.setenv("PGX_PERSIST", "0")
    b = VecPogema(gc, batch=B, auto_reset=True)
    import ctypes as C
    shape_a, shape_b = (C.c_int32 * 4)(), (C.c_int32 * 4)()
    a._lib.pgx_debug_launch_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    assert a._lib.pgx_debug_launch_shape(a._handle, shape_a) == 0 and a._lib.pgx_debug_launch_shape(b._handle, shape_b) == 0
    assert list(shape_a)[:3] == [1, 256, 3] and shape_b[0] == 0, (list(shape_a), list(shape_b))
    oa, _ = a.reset(seed=4)
    ob, _ = b.reset(seed=4)
    assert torch.equal(oa, ob)
    acts = torch.randint(0, 5, (40, B, 64), device="cuda", dtype=torch.int8)
    for t in range(40):
        ra, rb = a.step(acts[t]), b.step(acts[t])
        for x, y in zip(ra[:4], rb[:4]):
            assert torch.equal(x, y), t
        assert torch.equal(ra[4]["is_active"], rb[4]["is_active"]) and torch.equal(ra[4]["metrics"], rb[4]["metrics"])
    assert torch.equal(a.observe(), b.observe())
    sa, sb = a.get_state(), b.get_state()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    a.close(); b.close()

"""The pinning pipeline, end to end, BEFORE the real package exists here: tools/gen_golden.py runs against a stand-in
`pogema` module built from the oracle (tests/standin_pogema), writes reference_*.npz into a TEMPORARY directory, and the
loader / comparison of tests/test_golden_reference.py is run on the result -- early termination, `elapsed`, `is_active`
and the lifelong special case included.  Nothing here says anything about parity with upstream (the stand-in IS the
oracle); it proves that the one-step pinning procedure of INTEGRATION.md section 5 works."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

import test_golden_reference as tgr
from util import assert_rollouts_equal, engine_rollout, oracle_rollout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN = os.path.join(ROOT, "tests", "standin_pogema")


def _generate(out_dir, limit):
    env = dict(os.environ, PYTHONPATH=STANDIN + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_golden.py"), "--out", str(out_dir),
                           "--limit", str(limit)], capture_output=True, text=True, env=env, timeout=900)


@pytest.fixture(scope="module")
def fixtures(tmp_path_factory):
    out = tmp_path_factory.mktemp("golden_standin")
    p = _generate(out, 27)  # geometry 0 (BASELINE configs[0]) x 3 collision systems x 3 on_target x 3 seeds
    assert p.returncode == 0, p.stderr
    files = sorted(glob.glob(os.path.join(str(out), "reference_*.npz")))
    assert len(files) == 27 and "wrote 27 fixtures" in p.stdout
    return files


def test_generator_refuses_to_write_standin_output_into_tests_golden():
    p = _generate(os.path.join(ROOT, "tests", "golden"), 1)
    assert p.returncode != 0 and "STAND-IN" in p.stderr
    assert not glob.glob(os.path.join(ROOT, "tests", "golden", "reference_*.npz"))


def test_fixture_format_and_early_termination(fixtures):
    seen_short = False
    for path in fixtures:
        z = np.load(path, allow_pickle=False)
        T, A = z["actions"].shape
        assert z["obs"].shape[:2] == (T, A) and z["agents_xy"].shape == (T, A, 2) and z["obs0"].shape[0] == A
        assert z["obstacles"].ndim == 2 and z["agents_xy0"].shape == (A, 2)
        done = z["terminated"][-1].all() or z["truncated"][-1].all()
        assert done, "every recorded episode runs to its end"
        seen_short = seen_short or T < int(z["max_episode_steps"])
    assert seen_short, "at least one episode must end before the time limit (exercises the action-stream truncation)"


def test_oracle_passes_the_loaded_fixtures(fixtures):
    """Includes the lifelong cases: the stand-in's target stream is seeded differently from the rollout under test
    (as the real package's numpy stream will be), so the target replay of compare_with_fixture is what makes them pass."""
    lifelong_with_goals = 0
    for path in fixtures:
        tgr.compare_with_fixture(oracle_rollout, path)
        z = np.load(path, allow_pickle=False)
        lifelong_with_goals += str(z["on_target"]) == "restart" and float(z["rewards"].sum()) > 0
    assert lifelong_with_goals >= 2, "the replay path must actually be exercised (lifelong fixtures with goals reached)"


def test_lifelong_replay_is_needed(fixtures):
    """Without the replay a lifelong fixture with a different target stream must NOT match (the replay has teeth)."""
    for path in fixtures:
        z = np.load(path, allow_pickle=False)
        if str(z["on_target"]) != "restart" or float(z["rewards"][:-1].sum()) == 0 or "_s0" in path:
            continue
        obstacles, agents, targets, actions, ref, kw = tgr._load(path)
        got = oracle_rollout(obstacles, agents, targets, actions, **kw)
        assert not np.array_equal(got["targets_xy"], ref["targets_xy"])
        return
    pytest.skip("no lifelong fixture with an early goal among the generated ones")


def test_comparison_detects_a_wrong_fixture(fixtures, tmp_path):
    """A fixture whose expected positions are off by one step must FAIL the comparison (the check has teeth)."""
    z = dict(np.load(fixtures[0], allow_pickle=False))
    z["agents_xy"] = np.roll(z["agents_xy"], 1, axis=0)
    bad = tmp_path / "reference_bad.npz"
    np.savez_compressed(bad, **z)
    with pytest.raises(AssertionError):
        tgr.compare_with_fixture(oracle_rollout, str(bad))


def test_generator_pin_on_the_loaded_fixtures(fixtures, tmp_path):
    """The instance-generator pin: host build of the generator kernel == the recorded initial state for every fixture,
    and it has teeth (a fixture from another seed fails)."""
    from pogema_amd.nprng import np_generate_host
    for path in fixtures:
        assert tgr.compare_generator_with_fixture(np_generate_host, path)
    z = dict(np.load(fixtures[0], allow_pickle=False))
    z["grid_seed"] = np.asarray(int(z["grid_seed"]) + 1)
    bad = tmp_path / "reference_bad_seed.npz"
    np.savez_compressed(bad, **z)
    with pytest.raises(AssertionError):
        tgr.compare_generator_with_fixture(np_generate_host, str(bad))


@pytest.mark.gpu
def test_device_generator_pin_on_the_loaded_fixtures(fixtures):
    from pogema_amd.nprng import np_generate
    for path in fixtures:
        assert tgr.compare_generator_with_fixture(
            lambda *a: tuple(v.cpu().numpy() for v in np_generate(*a)), path)


@pytest.mark.gpu
def test_engine_passes_the_loaded_fixtures(fixtures):
    for path in fixtures:
        tgr.compare_with_fixture(engine_rollout, path)


def _pin(out_dir, standin_semantics="", pin_file=None):
    env = dict(os.environ, PYTHONPATH=STANDIN + os.pathsep + os.environ.get("PYTHONPATH", ""),
               PGX_STANDIN_SEMANTICS=standin_semantics)
    env.pop("PGX_SEMANTICS", None)
    env.pop("PGX_PINNED_SEMANTICS_FILE", None)
    extra = ["--pin-file", str(pin_file)] if pin_file else []
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "pin_reference.sh"), "--out", str(out_dir), "--geoms", "3,0",
                        "--limit", "36"] + extra, capture_output=True, text=True, env=env, timeout=1500)
    import json
    return p, json.load(open(os.path.join(str(out_dir), "pin_report.json")))


def test_pin_reference_script_finds_the_standins_switches(tmp_path):
    """tools/pin_reference.sh end to end: generate -> brute-force the 2^4 switch positions over the fixtures
    -> default-semantics tests.  The stand-in's switches are the product defaults, so everything passes and every switch a
    rollout can show is DETERMINED by the fixtures (the collision-dense geometry exercises the soft-collision ones), the
    out-of-range-action one by the probe file."""
    p, rep = _pin(tmp_path / "default")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert rep["fixtures"] == 36 and rep["combinations_tried"] == 16 and rep["product_default_passes"] is True
    sw = rep["per_switch"]
    assert sw["soft_vertex"] == {"determined": "lowest_index"} and sw["soft_occupancy"] == {"determined": "index_order"}
    assert sw["coop_reward"] == {"determined": "all_solved"} and sw["bad_action"] == {"determined_by_probe": "noop"}
    assert rep["probes"]["standin"] is True and rep["probes"]["grid_config_defaults"]["collision_system"] == "priority"
    z = np.load(sorted(glob.glob(os.path.join(str(tmp_path / "default"), "reference_12x40_soft_*.npz")))[0], allow_pickle=False)
    assert z["positions"].shape[0] == z["actions"].shape[0] and z["positions0"].ndim == 2 and "metrics_names" in z.files


def test_pin_reference_script_names_the_flip_for_a_different_reference(tmp_path):
    """A 'reference' whose low-confidence details are all the OTHER way round (stand-in with non-default switches): the
    default-semantics tests fail, and the report names exactly the switches to flip -- from the fixtures alone."""
    p, rep = _pin(tmp_path / "flipped", "soft_vertex_rule=all_stay,soft_occupancy=exact,coop_reward=per_agent,bad_action=flag")
    assert p.returncode != 0 and rep["product_default_passes"] is False
    sw = rep["per_switch"]
    assert sw["soft_vertex"] == {"determined": "all_stay"} and sw["soft_occupancy"] == {"determined": "exact"}
    assert sw["coop_reward"] == {"determined": "per_agent"} and sw["bad_action"] == {"determined_by_probe": "flag"}
    assert {"soft_vertex": "all_stay", "soft_occupancy": "exact", "coop_reward": "per_agent", "bad_action": "noop"} in rep["passing"]


def test_pin_reference_script_flips_the_defaults_by_data(tmp_path, monkeypatch):
    """ADVICE r4 / SURVEY 8(f1): with a pin file the same 'different reference' ends GREEN in one command -- the switch
    positions its fixtures demand are written as the product's pinned defaults, the default-semantics tests then run under
    them and pass, and `Semantics.from_env()` (what every VecPogema built without an explicit `semantics=` uses) follows
    the file.  Precedence: PGX_SEMANTICS > pinned file > built-in recollections; a malformed file is an error."""
    import json
    from pogema_amd import semantics as S
    pin = tmp_path / "pinned_semantics.json"
    p, rep = _pin(tmp_path / "flipped", "soft_vertex_rule=all_stay,soft_occupancy=exact,coop_reward=per_agent,bad_action=flag", pin_file=pin)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert rep["product_default_passes"] is False and rep["pin_file"] == str(pin)
    data = json.load(open(pin))
    want = {"soft_vertex": "all_stay", "soft_occupancy": "exact", "coop_reward": "per_agent", "bad_action": "flag"}
    assert data["switches"] == want and data["differs_from_recalled_defaults"] == want and data["fixtures"] == 36
    assert not os.path.exists(os.path.join(ROOT, "pogema_amd", "pinned_semantics.json")), "a rehearsal must not pin the product"
    monkeypatch.delenv("PGX_SEMANTICS", raising=False)
    monkeypatch.delenv("PGX_PINNED_SEMANTICS_FILE", raising=False)
    assert S.Semantics.from_env() == S.Semantics() and S.pinned_source() is None
    monkeypatch.setenv("PGX_PINNED_SEMANTICS_FILE", str(pin))
    sem = S.Semantics.from_env()
    assert (sem.soft_vertex, sem.soft_occupancy, sem.coop_reward, sem.bad_action) == ("all_stay", "exact", "per_agent", "flag")
    assert sem.lifelong_rng == "build" and S.pinned_source() == str(pin)
    monkeypatch.setenv("PGX_SEMANTICS", "soft_occupancy=index_order")
    assert S.Semantics.from_env().soft_occupancy == "index_order" and S.Semantics.from_env().soft_vertex == "all_stay"
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps({"switches": {"soft_vertex": "sometimes"}}))
    monkeypatch.setenv("PGX_PINNED_SEMANTICS_FILE", str(bad))
    with pytest.raises(ValueError, match="not a known semantics switch"):
        S.Semantics.from_env()


STANDIN_SRC = os.path.join(ROOT, "tests", "standin_pogema_src")


def _grid_fixtures(tmp_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_golden_grid.py"), "--ref", STANDIN_SRC, "--out", str(tmp_path),
                        "--limit", "10"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    files = sorted(glob.glob(os.path.join(str(tmp_path), "reference_grid_*.npz")))
    assert len(files) == 10
    return files


def test_grid_layer_fixtures_without_importing_the_package(tmp_path):
    """a source tree whose `pogema/__init__.py` cannot be imported (no gymnasium) still yields fixtures
    of the grid layer -- tools/gen_golden_grid.py loads grid_config / generator / grid under a bare `pogema` namespace.
    The stand-in tree's __init__ raises ImportError, so the rehearsal proves the bypass; the fixtures pin rows A1-A3 and
    A9-A11 (oracle), the border ring, and the numpy-stream instance generator."""
    with pytest.raises(ImportError):
        import importlib.util
        spec = importlib.util.spec_from_file_location("pogema_standin_src_init", os.path.join(STANDIN_SRC, "pogema", "__init__.py"))
        spec.loader.exec_module(importlib.util.module_from_spec(spec))
    files = _grid_fixtures(tmp_path)
    from pogema_amd.nprng import np_generate_host
    for path in files:
        tgr.compare_grid_fixture(oracle_rollout, path)
        tgr.compare_border_with_grid_fixture(path)
        assert tgr.compare_generator_with_fixture(np_generate_host, path)
    # a corrupted fixture is caught: one agent one cell off after the third step
    z = dict(np.load(files[3], allow_pickle=False))
    z["agents_xy"] = z["agents_xy"].copy()
    z["agents_xy"][2, 0, 0] += 1
    bad = tmp_path / "reference_grid_bad.npz"
    np.savez_compressed(bad, **z)
    with pytest.raises(AssertionError):
        tgr.compare_grid_fixture(oracle_rollout, str(bad))
    # the generator refuses to put stand-in output into tests/golden, and says what is missing for a tree without grid.py
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_golden_grid.py"), "--ref", STANDIN_SRC], capture_output=True, text=True)
    assert p.returncode != 0 and "STAND-IN" in p.stderr
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_golden_grid.py"), "--ref", str(tmp_path), "--out", str(tmp_path)],
                       capture_output=True, text=True)
    assert p.returncode != 0 and "not found" in p.stderr


def test_pin_reference_script_grid_only_mode(tmp_path):
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "pin_reference.sh"), "--grid-only", "--ref", STANDIN_SRC, "--out",
                        str(tmp_path / "g"), "--limit", "6"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-1000:]
    assert "grid-layer tests rc=0" in p.stdout and "12 passed" in p.stdout  # 6 fixtures x (oracle + border ring, generator); engine: -m gpu


@pytest.mark.gpu
def test_engine_passes_the_grid_layer_fixtures(tmp_path):
    for path in _grid_fixtures(tmp_path):
        tgr.compare_grid_fixture(engine_rollout, path)

"""CPU (no GPU): the closed-form collision rules the HIP kernel evaluates (pgx_kernels.hip phase 2: all-pairs
okey/ckey sweep + pointer-doubling closure) restated in numpy-free Python and checked with hypothesis against the
LITERAL algorithms of the oracle (sequential `priority`, dict-based `block_both`, dict/recursion `soft`) on random
crowded scenarios.  This is the derivation DESIGN.md section 5 states, machine-checked on tens of thousands of cases;
tests/test_parity_gpu.py then checks the kernel itself."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle.pogema_oracle import MOVES, PogemaOracle

NOWHERE = ("nowhere",)
NOTHING = ("nothing",)


def closed_form_moves(obst_padded, cur, active, actions, collision, soft_vertex_rule="lowest_index"):
    """New padded cells per agent, following the kernel's rules (one 'lane' per agent)."""
    n = len(cur)
    mover = [active[i] and actions[i] != 0 for i in range(n)]
    dest = [(cur[i][0] + MOVES[actions[i]][0], cur[i][1] + MOVES[actions[i]][1]) for i in range(n)]
    blocked = [mover[i] and obst_padded[dest[i]] != 0 for i in range(n)]
    vis = [cur[i] if active[i] else NOWHERE for i in range(n)]
    claims = [(active[i] if collision == "block_both" else mover[i]) for i in range(n)]
    want = [dest[i] if claims[i] else NOTHING for i in range(n)]
    # the sweep: occupant of my destination, lower-index claimants of my destination
    occ = [next((j for j in range(n) if j != i and vis[j] == want[i]), -1) for i in range(n)]
    others = [[j for j in range(n) if j != i and want[j] == want[i]] for i in range(n)]
    stay = [False] * n
    nxt = [-1] * n
    for i in range(n):
        lower = [j for j in others[i] if j < i]
        c1 = max(lower) if lower else -1
        o = occ[i]
        if collision == "block_both":
            stay[i] = (not mover[i]) or blocked[i] or o >= 0 or len(others[i]) > 0
            continue
        nxt[i] = o if mover[i] else -1
        if collision == "priority":
            stay[i] = (not mover[i]) or blocked[i] or o > i or c1 > o
        else:
            stay[i] = (not mover[i]) or blocked[i] or bool(lower if soft_vertex_rule == "lowest_index" else others[i])
            if nxt[i] >= 0 and want[nxt[i]] == cur[i]:
                stay[i] = True  # edge swap
    if collision != "block_both":
        rounds = 1
        while (1 << rounds) < n:
            rounds += 1
        for _ in range(rounds):  # pointer doubling, synchronous rounds like the kernel
            if not any(nxt[i] >= 0 and not stay[i] for i in range(n)):
                break
            snap = [(stay[i], nxt[i]) for i in range(n)]
            for i in range(n):
                if nxt[i] >= 0:
                    s, nn = snap[nxt[i]]
                    stay[i] = stay[i] or s
                    nxt[i] = nn
    return [cur[i] if stay[i] else dest[i] for i in range(n)]


@st.composite
def scenarios(draw):
    h = draw(st.integers(2, 6))
    w = draw(st.integers(2, 6))
    cells = [(x, y) for x in range(h) for y in range(w)]
    n_obst = draw(st.integers(0, max(0, h * w // 4)))
    perm = draw(st.permutations(cells))
    obst_cells, free = perm[:n_obst], perm[n_obst:]
    n = draw(st.integers(1, min(len(free), 12)))
    starts = free[:n]
    targets = [draw(st.sampled_from(free)) for _ in range(n)]
    steps = draw(st.integers(1, 4))
    actions = [[draw(st.integers(0, 4)) for _ in range(n)] for _ in range(steps)]
    obstacles = np.zeros((h, w), np.uint8)
    for c in obst_cells:
        obstacles[c] = 1
    return obstacles, starts, targets, actions


@pytest.mark.parametrize("collision", ["priority", "block_both", "soft", "soft/all_stay"])
@pytest.mark.parametrize("on_target", ["finish", "nothing"])
@settings(max_examples=1500, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(sc=scenarios())
def test_closed_form_equals_literal_algorithm(collision, on_target, sc):
    obstacles, starts, targets, actions = sc
    collision, _, rule = collision.partition("/")  # 'soft/all_stay': the docs/SPEC.md Q1 alternative
    rule = rule or "lowest_index"
    env = PogemaOracle(obstacles, starts, targets, obs_radius=1, collision_system=collision, on_target=on_target,
                       max_episode_steps=1000, soft_vertex_rule=rule, soft_occupancy="exact")
    g = env.grid
    for acts in actions:
        cur = list(g.positions_xy)
        active = [bool(g.is_active[i]) for i in range(len(cur))]
        expect = closed_form_moves(g.obstacles, cur, active, acts, collision, rule)
        env.step(list(acts))
        assert list(g.positions_xy) == expect, (collision, cur, active, acts)
        # invariants every collision system keeps
        vis = [p for i, p in enumerate(g.positions_xy) if g.is_active[i]]
        assert len(set(vis)) == len(vis), "two visible agents share a cell"
        assert all(g.obstacles[p] == 0 for p in g.positions_xy), "an agent stands on an obstacle"
        assert all(abs(a[0] - b[0]) + abs(a[1] - b[1]) <= 1 for a, b in zip(cur, g.positions_xy))
        occ = np.zeros_like(g.positions)
        for p in vis:
            occ[p] = 1
        # (under `soft` this is the Q2 ALTERNATIVE, soft_occupancy='exact'; the default literal loop: next test)
        assert np.array_equal(occ, g.positions), "occupancy array == cells of the visible agents"
        if collision != "priority":  # no edge swaps under block_both / soft
            for i in range(len(cur)):
                for j in range(i + 1, len(cur)):
                    if active[i] and active[j] and cur[i] != cur[j]:
                        assert not (g.positions_xy[i] == cur[j] and g.positions_xy[j] == cur[i])


@pytest.mark.parametrize("rule", ["lowest_index", "all_stay"])
@settings(max_examples=1500, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(sc=scenarios())
def test_soft_occupancy_index_order_closed_form(rule, sc):
    """docs/SPEC.md Q2 default (soft_occupancy='index_order', the recalled literal): the kernel's closed form of the
    per-agent clear-old / set-new loop -- an agent is missing from the occupancy array iff it MOVED into a cell whose
    previous occupant has a HIGHER index, and a step recomputes that from scratch (every active agent takes its turn
    again, so last step's missing agents reappear unless they qualify anew) -- against the oracle's literal loop over
    its PERSISTENT array; positions themselves never depend on the switch."""
    obstacles, starts, targets, actions = sc
    kw = dict(obs_radius=1, collision_system="soft", on_target="nothing", max_episode_steps=1000, soft_vertex_rule=rule)
    env = PogemaOracle(obstacles, starts, targets, **kw)
    assert env.soft_occupancy == "index_order", "the default is the recalled literal"
    exact = PogemaOracle(obstacles, starts, targets, soft_occupancy="exact", **kw)
    g = env.grid
    for acts in actions:
        cur = list(g.positions_xy)
        env.step(list(acts))
        exact.step(list(acts))
        new = list(g.positions_xy)
        assert new == list(exact.grid.positions_xy)
        standing = {c: j for j, c in enumerate(cur)}
        expect = np.zeros_like(g.positions)
        for i, c in enumerate(new):
            ghost = c != cur[i] and standing.get(c, -1) > i
            if not ghost:
                expect[c] = 1
        assert np.array_equal(expect, g.positions), (cur, acts, new)

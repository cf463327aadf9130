"""Writes tests/golden/spec_vectors.json: HAND-DERIVED known-answer cases for the step path.

These are NOT outputs of the reference (which is not available in this container: /root/reference
holds only README.md) and NOT outputs of the oracle either -- every expectation below was worked out
by hand from SURVEY.md section 8a (rows A2..A13) and is stated with its reasoning in `why`, so the
oracle (tests/test_oracle.py) and the HIP engine (tests/test_spec_vectors_gpu.py) are both checked
against something neither of them produced.  When a real `pogema` checkout becomes importable,
tools/gen_golden.py replaces/extends these with true reference vectors.
"""
import json
import os

OPEN3 = [[0, 0, 0], [0, 0, 0], [0, 0, 0]]
NOOP, UP, DOWN, LEFT, RIGHT = 0, 1, 2, 3, 4
cases = []


def case(name, why, map_, agents, targets, actions, collision, expect, on_target="finish", r=1, max_steps=64,
         semantics=None):
    """`semantics`: non-default switches of pogema_amd.Semantics (docs/SPEC.md Q1 / Q4 / Q7) this case runs under."""
    tag = "" if not semantics else "/" + ",".join(f"{k}={v}" for k, v in sorted(semantics.items()))
    c = dict(name=f"{name}-{collision}{tag}", why=why, map=map_, agents_xy=agents, targets_xy=targets,
             actions=actions, collision_system=collision, on_target=on_target, obs_radius=r,
             max_episode_steps=max_steps, expect=expect)
    if semantics:
        c["semantics"] = semantics
    cases.append(c)


far = [[2, 2], [2, 0]]
# --- vertex conflict ------------------------------------------------------------------------------------
case("vertex", "A3: agents move in index order; 0 takes (0,1) first, 1 then finds it occupied",
     OPEN3, [[0, 0], [0, 2]], far, [[RIGHT, LEFT]], "priority",
     dict(agents_xy=[[[0, 1], [0, 2]]],
          obs0_agent0=[[[1, 1, 1], [1, 0, 0], [1, 0, 0]], [[0, 0, 0], [0, 1, 0], [0, 0, 0]], [[0, 0, 0], [0, 0, 0], [0, 0, 1]]]))
case("vertex", "A4: a destination claimed twice is blocked for both", OPEN3, [[0, 0], [0, 2]], far,
     [[RIGHT, LEFT]], "block_both", dict(agents_xy=[[[0, 0], [0, 2]]]))
case("vertex", "A5 literal algorithm: the reverse-index loop reverts agent 1 first (2 claimants), which REMOVES it "
     "from used_cells[(0,1)]; agent 0 is then the sole claimant and moves (open parity question 1 in DESIGN.md)",
     OPEN3, [[0, 0], [0, 2]], far, [[RIGHT, LEFT]], "soft", dict(agents_xy=[[[0, 1], [0, 2]]]))
# --- edge swap ------------------------------------------------------------------------------------------
for cs, why in (("priority", "A3: each finds the other's cell occupied at its turn"),
                ("block_both", "A4: every currently occupied cell is blocked"),
                ("soft", "A5: edge (swap) conflict reverts both")):
    case("swap", why, OPEN3, [[0, 0], [0, 1]], far, [[RIGHT, LEFT]], cs, dict(agents_xy=[[[0, 0], [0, 1]]]))
# --- following ------------------------------------------------------------------------------------------
case("follow_lower_leads", "A3: agent 0 vacates (0,1) before agent 1's turn, so 1 may enter it",
     OPEN3, [[0, 1], [0, 0]], far, [[RIGHT, RIGHT]], "priority", dict(agents_xy=[[[0, 2], [0, 1]]]))
case("follow_lower_leads", "A4: (0,1) is an agent's current cell -> blocked; the leader's own move is free",
     OPEN3, [[0, 1], [0, 0]], far, [[RIGHT, RIGHT]], "block_both", dict(agents_xy=[[[0, 2], [0, 0]]]))
case("follow_lower_leads", "A5: following is allowed", OPEN3, [[0, 1], [0, 0]], far, [[RIGHT, RIGHT]], "soft",
     dict(agents_xy=[[[0, 2], [0, 1]]]))
case("follow_higher_leads", "A3: at agent 0's turn agent 1 has not moved yet -> 0 is blocked; 1 then moves",
     OPEN3, [[0, 0], [0, 1]], far, [[RIGHT, RIGHT]], "priority", dict(agents_xy=[[[0, 0], [0, 2]]]))
case("follow_higher_leads", "A4: as above, follower blocked", OPEN3, [[0, 0], [0, 1]], far, [[RIGHT, RIGHT]],
     "block_both", dict(agents_xy=[[[0, 0], [0, 2]]]))
case("follow_higher_leads", "A5: order-independent following", OPEN3, [[0, 0], [0, 1]], far, [[RIGHT, RIGHT]],
     "soft", dict(agents_xy=[[[0, 1], [0, 2]]]))
# --- 2x2 rotation -----------------------------------------------------------------------------------------
rot_agents = [[0, 0], [0, 1], [1, 1], [1, 0]]
rot_targets = [[2, 2], [2, 1], [2, 0], [0, 2]]
rot_actions = [[RIGHT, DOWN, LEFT, UP]]
case("rotation", "A3: every destination is occupied at the mover's turn", OPEN3, rot_agents, rot_targets, rot_actions,
     "priority", dict(agents_xy=[rot_agents]))
case("rotation", "A4: all destinations are currently occupied cells", OPEN3, rot_agents, rot_targets, rot_actions,
     "block_both", dict(agents_xy=[rot_agents]))
case("rotation", "A5: cyclic rotation is allowed (one claimant per cell, no swap edge)", OPEN3, rot_agents, rot_targets,
     rot_actions, "soft", dict(agents_xy=[[[0, 1], [1, 1], [1, 0], [0, 0]]]))
# --- obstacles, border ring, cascade --------------------------------------------------------------------------
ROW = [[0, 0, 0, 1]]
for cs in ("priority", "block_both", "soft"):
    case("blocked_chain", "front agent faces an obstacle; under every system nobody behind it can advance "
         "(A3: occupied at turn; A4: occupied cells blocked; A5: revert cascades down the chain)",
         ROW, [[0, 0], [0, 1], [0, 2]], [[0, 2], [0, 0], [0, 1]], [[RIGHT, RIGHT, RIGHT]], cs,
         dict(agents_xy=[[[0, 0], [0, 1], [0, 2]]]))
    case("border", "A1: the wall ring around the map blocks moves off the map", OPEN3, [[0, 0], [2, 2]], [[1, 1], [1, 0]],
         [[UP, RIGHT], [LEFT, DOWN]], cs, dict(agents_xy=[[[0, 0], [2, 2]], [[0, 0], [2, 2]]]))
case("stayer_wins", "A5: a noop agent keeps its cell; the mover into it is reverted and so is the one following it",
     OPEN3, [[0, 0], [0, 1], [0, 2]], [[2, 2], [2, 1], [2, 0]], [[NOOP, LEFT, LEFT]], "soft",
     dict(agents_xy=[[[0, 0], [0, 1], [0, 2]]]))
case("three_way", "A5 literal: three movers claim (1,1); reverse-index reverts 2 then 1, agent 0 remains sole claimant; "
     "agent 3 follows the reverted agent 2 and is reverted by the cascade",
     [[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]], [[0, 1], [1, 0], [1, 2], [1, 3]], [[2, 3], [2, 2], [2, 1], [2, 0]],
     [[DOWN, RIGHT, LEFT, LEFT]], "soft", dict(agents_xy=[[[1, 1], [1, 0], [1, 2], [1, 3]]]))
# --- goals / hide / time limit / cooperative finish ------------------------------------------------------------
case("finish_hide", "A6: reward once on arrival, terminated stays true, the finished agent is hidden so another agent "
     "may enter its cell; A2 noop keeps position",
     OPEN3, [[0, 0], [0, 2]], [[0, 1], [2, 2]], [[RIGHT, NOOP], [RIGHT, LEFT]], "priority",
     dict(agents_xy=[[[0, 1], [0, 2]], [[0, 1], [0, 1]]], rewards=[[1.0, 0.0], [0.0, 0.0]],
          terminated=[[1, 0], [1, 0]], is_active=[[0, 1], [0, 1]], truncated=[[0, 0], [0, 0]]))
case("time_limit", "A13: truncated for all agents once elapsed >= max_episode_steps", OPEN3, [[0, 0], [2, 2]],
     [[1, 1], [1, 0]], [[NOOP, NOOP], [NOOP, NOOP], [NOOP, NOOP]], "priority",
     dict(agents_xy=[[[0, 0], [2, 2]]] * 3, truncated=[[0, 0], [1, 1], [1, 1]], terminated=[[0, 0]] * 3), max_steps=2)
case("coop_finish", "A8: nothing happens on a goal until ALL agents stand on theirs; then everyone gets 1.0 and terminates",
     OPEN3, [[0, 0], [2, 2]], [[0, 1], [2, 0]], [[RIGHT, LEFT], [NOOP, LEFT]], "priority",
     dict(agents_xy=[[[0, 1], [2, 1]], [[0, 1], [2, 0]]], rewards=[[0.0, 0.0], [1.0, 1.0]], terminated=[[0, 0], [1, 1]],
          is_active=[[1, 1], [1, 1]]), on_target="nothing")

# --- the switches for the low-confidence recollections (docs/SPEC.md Q1, Q4, Q7): both values of each ---------
ALL_STAY = {"soft_vertex": "all_stay"}
case("vertex", "Q1 alternative: both claimants of the free cell (0,1) stay", OPEN3, [[0, 0], [0, 2]], far,
     [[RIGHT, LEFT]], "soft", dict(agents_xy=[[[0, 0], [0, 2]]]), semantics=ALL_STAY)
case("three_way", "Q1 alternative: all three claimants of (1,1) stay; agent 3 moves into the cell of agent 2, which "
     "stays, so 3 stays as well", [[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]], [[0, 1], [1, 0], [1, 2], [1, 3]],
     [[2, 3], [2, 2], [2, 1], [2, 0]], [[DOWN, RIGHT, LEFT, LEFT]], "soft",
     dict(agents_xy=[[[0, 1], [1, 0], [1, 2], [1, 3]]]), semantics=ALL_STAY)
vac_agents = [[1, 1], [1, 0], [1, 2]]
vac_targets = [[0, 0], [0, 2], [2, 0]]
case("vacated_contest", "Q1 literal: agent 0 leaves (1,1) downwards; 1 and 2 both claim the vacated cell; the "
     "reverse-index loop reverts 2, agent 1 is then the only claimant and follows into (1,1)",
     OPEN3, vac_agents, vac_targets, [[DOWN, RIGHT, LEFT]], "soft", dict(agents_xy=[[[2, 1], [1, 1], [1, 2]]]))
case("vacated_contest", "Q1 alternative: the vacated cell is contested, so both claimants stay; the leaver moves",
     OPEN3, vac_agents, vac_targets, [[DOWN, RIGHT, LEFT]], "soft", dict(agents_xy=[[[2, 1], [1, 0], [1, 2]]]),
     semantics=ALL_STAY)
ROW5 = [[0, 0, 0, 0, 0]]
cas_agents = [[0, 1], [0, 3], [0, 0]]
cas_targets = [[0, 4], [0, 0], [0, 3]]
case("contest_then_follow", "Q1 literal: 0 and 1 claim (0,2): index 0 wins and moves, 1 stays; 2 follows into the cell "
     "0 vacated", ROW5, cas_agents, cas_targets, [[RIGHT, LEFT, RIGHT]], "soft",
     dict(agents_xy=[[[0, 2], [0, 3], [0, 1]]]))
case("contest_then_follow", "Q1 alternative: 0 and 1 both stay; 2 wants the cell of 0, which stays, so 2 stays too "
     "(second round of the fixed point)", ROW5, cas_agents, cas_targets, [[RIGHT, LEFT, RIGHT]], "soft",
     dict(agents_xy=[[[0, 1], [0, 3], [0, 0]]]), semantics=ALL_STAY)
case("coop_finish", "Q4 alternative: each agent is paid 1.0 in every step it stands on its own goal; the episode still "
     "terminates only when all do", OPEN3, [[0, 0], [2, 2]], [[0, 1], [2, 0]], [[RIGHT, LEFT], [NOOP, LEFT]], "priority",
     dict(agents_xy=[[[0, 1], [2, 1]], [[0, 1], [2, 0]]], rewards=[[1.0, 0.0], [1.0, 1.0]], terminated=[[0, 0], [1, 1]],
          is_active=[[1, 1], [1, 1]]), on_target="nothing", semantics={"coop_reward": "per_agent"})
for cs in ("priority", "block_both", "soft"):
    case("bad_action", "Q7 default: actions outside 0..4 (too large or negative) do nothing", OPEN3, [[0, 0], [2, 2]],
         far, [[7, -3], [RIGHT, 5]], cs, dict(agents_xy=[[[0, 0], [2, 2]], [[0, 1], [2, 2]]]))
    case("bad_action", "Q7 alternative: an out-of-range action of an ACTIVE agent raises the reference's IndexError",
         OPEN3, [[0, 0], [2, 2]], far, [[RIGHT, 5]], cs, dict(raises="IndexError"), semantics={"bad_action": "flag"})
case("bad_action_inactive", "Q7 alternative: a finished (hidden) agent's action is never looked at (the reference guards "
     "every MOVES[action] with is_active), so its garbage action raises nothing",
     OPEN3, [[0, 0], [0, 2]], [[0, 1], [2, 2]], [[RIGHT, NOOP], [9, DOWN]], "priority",
     dict(agents_xy=[[[0, 1], [0, 2]], [[0, 1], [1, 2]]], is_active=[[0, 1], [0, 1]]), semantics={"bad_action": "flag"})

# --- Q2: the occupancy array after a `soft` step ------------------------------------------------------------------
both_visible = [[[0, 0, 0], [0, 1, 1], [0, 0, 0]], [[0, 0, 0], [1, 1, 0], [0, 0, 0]]]  # agents planes of agent 0 / agent 1
case("follow_occupancy", "Q2 alternative (soft_occupancy='exact'): agent 0 (0,0)->(0,1) follows agent 1 (0,1)->(0,2); afterwards "
     "the occupancy array is exactly the two agents' cells, so each sees itself in the centre and the other next to it; a noop "
     "step changes nothing", OPEN3, [[0, 0], [0, 1]], far, [[RIGHT, RIGHT], [NOOP, NOOP]], "soft",
     dict(agents_xy=[[[0, 1], [0, 2]], [[0, 1], [0, 2]]], agents_plane=[both_visible, both_visible]),
     semantics={"soft_occupancy": "exact"})
case("follow_occupancy", "Q2 default, literal move_without_checks in index order: agent 0 clears (0,0) and sets (0,1); then "
     "agent 1 clears ITS old cell (0,1) -- where agent 0 now stands -- and sets (0,2): agent 0 is missing from the occupancy "
     "array (its own centre cell and agent 1's left neighbour read 0); in the next step agent 0's own turn (a noop: clear, "
     "set) puts it back", OPEN3, [[0, 0], [0, 1]], far, [[RIGHT, RIGHT], [NOOP, NOOP]], "soft",
     dict(agents_xy=[[[0, 1], [0, 2]], [[0, 1], [0, 2]]],
          agents_plane=[[[[0, 0, 0], [0, 0, 1], [0, 0, 0]], [[0, 0, 0], [0, 1, 0], [0, 0, 0]]], both_visible]))
case("follow_occupancy", "Q2 default, lower index leads: agent 0 (0,1)->(0,2) sets (0,2) first, agent 1 (0,0)->(0,1) "
     "then clears (0,0) and sets (0,1) -- nobody is cleared afterwards: identical with the alternative",
     OPEN3, [[0, 1], [0, 0]], far, [[RIGHT, RIGHT]], "soft",
     dict(agents_xy=[[[0, 2], [0, 1]]],
          agents_plane=[[[[0, 0, 0], [1, 1, 0], [0, 0, 0]], [[0, 0, 0], [0, 1, 1], [0, 0, 0]]]]))
case("follow_occupancy", "Q2 default, three in a row moving right with the LAST index in front (agents 0,1 follow 2): agent 0 "
     "(0,0)->(0,1) sets (0,1), agent 1 (0,1)->(0,2) clears (0,1) and sets (0,2), agent 2 (0,2)->(0,3) clears (0,2) and sets "
     "(0,3): only agent 2 is left in the occupancy array; agent 0 sees nothing in its 3x3 window, agent 1 sees agent 2 to its "
     "right, agent 2 sees itself", [[0, 0, 0, 0, 0]], [[0, 0], [0, 1], [0, 2]], [[0, 4], [0, 4], [0, 4]], [[RIGHT, RIGHT, RIGHT]], "soft",
     dict(agents_xy=[[[0, 1], [0, 2], [0, 3]]],
          agents_plane=[[[[0, 0, 0], [0, 0, 0], [0, 0, 0]], [[0, 0, 0], [0, 0, 1], [0, 0, 0]], [[0, 0, 0], [0, 1, 0], [0, 0, 0]]]]))

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "spec_vectors.json")
with open(out, "w") as f:
    json.dump(dict(provenance="hand-derived from SURVEY.md section 8a; NOT reference outputs (reference unavailable)",
                   cases=cases), f, indent=1)
print(len(cases), "cases ->", out)

"""CPU (hipcc cross-compiles gfx950 here): register / scratch / occupancy budget of the step kernels.

The single-wave step kernels must keep 8 waves per SIMD resident (residency matters more than
anything else) -- at most 64 VGPRs and 80-odd SGPRs -- and must not spill: an innocent extra branch inside the
specialised row-mask instances once cost 430-456 bytes of scratch per lane without any test noticing (round 4, the
16-bit observation formats).  `tools/resource_usage.py` reads hipcc's -Rpass-analysis=kernel-resource-usage remarks."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hipcc():
    import shutil
    return shutil.which("hipcc") or next((c for c in ("/opt/rocm/bin/hipcc",) if os.path.exists(c)), None)


@pytest.fixture(scope="module")
def usage():
    if _hipcc() is None:  # an environment reason, not a regression
        pytest.skip("no hipcc on this box: the gfx950 resource remarks cannot be produced")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "resource_usage.py"), "kernel"], capture_output=True, text=True,
                       cwd=ROOT, timeout=900)
    rows = {}
    for ln in p.stdout.splitlines():
        m = re.match(r"(\w+)<G=(\d+),MW=(\d),P16=(\d)>\s+sgpr\s+(\d+) vgpr\s+(\d+) scratch\s+(\d+) occ (\d+)", ln)
        # (the large-map instances are listed as step_kernel_big / rollout_kernel_big: names of their own)
        if m:
            rows[(m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)))] = tuple(int(m.group(k)) for k in (5, 6, 7, 8))
    if not rows and ("No such file" in p.stderr or "not found" in p.stderr or "cannot find ROCm" in p.stderr):
        pytest.skip("hipcc cannot cross-compile gfx950 here: " + p.stderr[-300:])
    assert len(rows) == 42, p.stdout[-2000:] + p.stderr[-2000:]  # 16 step + 16 rollout + 2 + 2 large-map + 6 resolver / streamer pairs
    return rows


# (sgpr, vgpr, scratch bytes per lane, waves per SIMD) of every step_kernel<G, MW, P16> instance as of the end of round 4
# (commit c95e51e).  Scope is frozen: the float32 observation path -- P16 = 1 are its specialised
# row-mask instances for windows up to 16 cells (obs_radius 3 / 5 / 7 = W 7 / 11 / 15), P16 = 0 the run-time-W form that
# also carries the light formats -- must not pay for anything added around it.  A change here is either a deliberate
# kernel change (update the table in the same commit, say why) or a regression.
# Round 6: unchanged -- the rollout form of step_body's load phase, the large-map layout and the resolver / streamer pair are
# all `if constexpr` branches the single-step instances do not see (one slip -- `p.mode` read in front of the global loads --
# showed as one VGPR less in the G = 1 instances and 0.5-1 % per launch; profiles/r6/step_ab_r5_vs_r6.txt).
FROZEN_STEP_KERNELS = {
    (64, 1, 1): (84, 60, 0, 8), (64, 1, 0): (100, 44, 0, 8),
    (1, 0, 1): (78, 60, 0, 8), (1, 0, 0): (78, 47, 0, 8),
    (2, 0, 1): (78, 60, 0, 8), (2, 0, 0): (78, 49, 0, 8),
    (4, 0, 1): (78, 60, 0, 8), (4, 0, 0): (78, 49, 0, 8),
    (8, 0, 1): (78, 60, 0, 8), (8, 0, 0): (78, 49, 0, 8),
    (16, 0, 1): (78, 60, 0, 8), (16, 0, 0): (78, 49, 0, 8),
    (32, 0, 1): (78, 60, 0, 8), (32, 0, 0): (78, 59, 0, 8),
    (64, 0, 1): (78, 59, 0, 8), (64, 0, 0): (78, 41, 0, 8),
}


def test_step_kernel_budgets_are_frozen(usage):
    got = {k[1:]: v for k, v in usage.items() if k[0] == "step_kernel"}
    assert got == FROZEN_STEP_KERNELS, {k: (got.get(k), FROZEN_STEP_KERNELS.get(k)) for k in set(got) | set(FROZEN_STEP_KERNELS)
                                        if got.get(k) != FROZEN_STEP_KERNELS.get(k)}


def test_step_kernels_do_not_spill_and_keep_eight_waves_per_simd(usage):
    for (name, G, mw, p16), (sgpr, vgpr, scratch, occ) in usage.items():
        if name != "step_kernel":
            continue
        assert scratch == 0, f"step_kernel<{G},{mw},{p16}> spills {scratch} bytes per lane"
        assert occ == 8 and vgpr <= 64, f"step_kernel<{G},{mw},{p16}>: {vgpr} VGPRs, {occ} waves per SIMD"
        if not mw:
            assert sgpr <= 80, f"single-wave step_kernel<{G},{mw},{p16}>: {sgpr} SGPRs (> 80 costs the eighth wave per SIMD)"


def test_rollout_kernels_stay_within_their_known_budget(usage):
    """Round 6: the rollout kernels are the low-occupancy instances -- `__launch_bounds__(64, 4)`, up to
    128 VGPRs -- because they keep the agent / env state, the action block in flight and the specialised row / stream code
    in registers across the K iterations and never wait for a store; a loop that does not wait does not need eight waves
    per SIMD (profiles/r6/rollout_ab_*.txt: faster than the 64-VGPR kernels of round 5 on every BASELINE config).  What
    must hold: no scratch at all (the round-5 kernels spilled 8-76 bytes per lane) and at least four waves per SIMD."""
    seen = 0
    for (name, G, mw, p16), (sgpr, vgpr, scratch, occ) in usage.items():
        if name in ("rollout_kernel_big", "rollout_kernel_pair"):
            assert scratch == 0 and occ >= 4 and vgpr <= 128, (name, G, vgpr, scratch, occ)
        if name != "rollout_kernel":
            continue
        seen += 1
        assert scratch == 0, f"rollout_kernel<{G},{mw},{p16}> spills {scratch} bytes per lane"
        assert occ >= 4 and vgpr <= 128, f"rollout_kernel<{G},{mw},{p16}>: {vgpr} VGPRs, {occ} waves per SIMD"
    assert seen == 16


def test_large_map_instances_do_not_spill(usage):
    """Round 6: step_kernel<64, true, P16, BIG> -- only the occupancy bitmap in LDS, obstacles through the L2.  One such
    workgroup owns a CU's LDS anyway (>= 80 KB of it), so occupancy is not the point; scratch is."""
    big = {k: v for k, v in usage.items() if k[0] == "step_kernel_big"}
    assert sorted(k[1:] for k in big) == [(64, 1, 0), (64, 1, 1)]
    for k, (sgpr, vgpr, scratch, occ) in big.items():
        assert scratch == 0 and vgpr <= 128 and occ >= 4, (k, vgpr, scratch, occ)

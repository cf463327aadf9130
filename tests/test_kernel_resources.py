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


@pytest.fixture(scope="module")
def usage():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "resource_usage.py"), "kernel"], capture_output=True, text=True,
                       cwd=ROOT, timeout=900)
    rows = {}
    for ln in p.stdout.splitlines():
        m = re.match(r"(\w+)<G=(\d+),MW=(\d),P16=(\d)>\s+sgpr\s+(\d+) vgpr\s+(\d+) scratch\s+(\d+) occ (\d+)", ln)
        if m:
            rows[(m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)))] = tuple(int(m.group(k)) for k in (5, 6, 7, 8))
    assert len(rows) == 32, p.stdout[-2000:] + p.stderr[-2000:]
    return rows


def test_step_kernels_do_not_spill_and_keep_eight_waves_per_simd(usage):
    for (name, G, mw, p16), (sgpr, vgpr, scratch, occ) in usage.items():
        if name != "step_kernel":
            continue
        assert scratch == 0, f"step_kernel<{G},{mw},{p16}> spills {scratch} bytes per lane"
        assert occ == 8 and vgpr <= 64, f"step_kernel<{G},{mw},{p16}>: {vgpr} VGPRs, {occ} waves per SIMD"
        if not mw:
            assert sgpr <= 80, f"single-wave step_kernel<{G},{mw},{p16}>: {sgpr} SGPRs (> 80 costs the eighth wave per SIMD)"


def test_rollout_kernels_stay_within_their_known_budget(usage):
    """The rollout kernels sit at the 64-VGPR cap with a little scratch: pin the order of
    magnitude so that a regression like the one above shows."""
    for (name, G, mw, p16), (sgpr, vgpr, scratch, occ) in usage.items():
        if name != "rollout_kernel":
            continue
        assert scratch <= 128, f"rollout_kernel<{G},{mw},{p16}> spills {scratch} bytes per lane"
        assert occ >= 7

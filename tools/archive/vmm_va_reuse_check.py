"""Diagnostic (round 3): is it safe to RE-USE a reserved address range for new physical memory (hipMemUnmap ->
hipMemMap of a fresh handle at the same address, the range itself never given back)?

Round 2 found that hipMemAddressFree + hipMemAddressReserve at the same address can keep the OLD translation alive on
ROCm 7.2 (profiles/r2/vmm_va_reuse_fault.txt), so the pool stopped freeing ranges and leaked them.  Round 3 keeps a
process-wide free list of released ranges (pgx_buffers.hip: g_va_free) and maps later pools into them.  This script
builds and destroys pools of the same sizes over and over (every pool after the first lands in re-used ranges), writes
and verifies every byte, and keeps canary tensors from torch's allocator alive in between: a stale translation would
show as a wrong read-back, a corrupted canary or a GPU memory fault.

    python tools/vmm_va_reuse_check.py [reps]        (PGX_VA_REUSE=0: the leaking behaviour, for comparison)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pogema_amd import _lib  # noqa: E402
from pogema_amd.buffers import ZoneBuffers  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
lib = _lib.load()
shape = (4096, 256, 3, 15, 15)  # configs[4]'s observation tensor, 2.8 GB
canaries = []
reserved = []
for rep in range(reps):
    for count in (2, 3):
        pool = ZoneBuffers(shape, torch.float32, "cuda:0", count=count)
        reserved.append(int(lib.pgx_buffers_va_reserved()))
        for i, t in enumerate(pool.tensors):
            t.fill_(float(rep * 10 + i))
        torch.cuda.synchronize()
        for i, t in enumerate(pool.tensors):
            v = float(rep * 10 + i)
            flat = t.view(-1)
            assert float(flat[0]) == v and float(flat[-1]) == v and bool((flat == v).all()), (rep, count, i)
        print(f"rep {rep} count {count}: spread {pool.info['spread']} after {pool.info['spacer_gib']:.0f} GiB, "
              f"ptr {hex(pool.ring.data_ptr())}, address space reserved so far {reserved[-1] / 2**30:.1f} GiB", flush=True)
        del pool, t, flat
        torch.cuda.synchronize()
        # memory the driver may hand out from the pages just released: must stay intact while later pools come and go
        c = torch.full(((1 << 30) // 4,), float(len(canaries) + 1), dtype=torch.float32, device="cuda")
        canaries.append(c)
        if len(canaries) > 8:
            old = canaries.pop(0)
            del old
for k, c in enumerate(canaries):
    assert bool((c == c[0]).all()) and float(c[0]) >= 1.0, f"canary {k} corrupted"
grow = [b - a for a, b in zip(reserved, reserved[1:])]
print("address space reserved after each pool [GiB]:", [round(r / 2**30, 1) for r in reserved])
print("OK: every byte answered, canaries intact; growth after the second repetition:",
      round(sum(grow[4:]) / 2**30, 1), "GiB")

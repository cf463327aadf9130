"""One-off probe (round 5): does torch see hipHostRegister'ed shared memory as pinned, and what does D2H sustain?"""
import mmap, os, time, ctypes
import torch
n = 1 << 30
path = f"/dev/shm/pgx_probe_{os.getpid()}"
fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
os.ftruncate(fd, n)
mm = mmap.mmap(fd, n)
os.unlink(path)
host = torch.frombuffer(mm, dtype=torch.uint8)
host.fill_(0)
print("before register is_pinned:", host.is_pinned())
rc = torch.cuda.cudart().cudaHostRegister(host.data_ptr(), n, 0)
print("cudaHostRegister rc:", rc, "is_pinned after:", host.is_pinned())
dev = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
side = torch.cuda.Stream()
for name, dst in (("registered shm", host), ("torch pinned", pin)):
    for size in (4 << 20, 64 << 20, 1 << 30):
        with torch.cuda.stream(side):
            dst[:size].copy_(dev[:size], non_blocking=True)
            side.synchronize()
            t0 = time.perf_counter()
            reps = 20 if size < (1 << 30) else 4
            for _ in range(reps):
                dst[:size].copy_(dev[:size], non_blocking=True)
            t_issue = time.perf_counter() - t0
            side.synchronize()
            dt = time.perf_counter() - t0
        print(f"{name}: {size >> 20} MiB x {reps}: {size * reps / dt / 1e9:.1f} GB/s, issue {t_issue / reps * 1e6:.1f} us/copy, ok={bool((dst[:size] == dev[:size].cpu()).all()) if size <= (64<<20) else 'n/a'}")
# ctypes path
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
size = 4 << 20
t0 = time.perf_counter()
for _ in range(50):
    rc = hip.hipMemcpyAsync(host.data_ptr(), dev.data_ptr(), size, 2, side.cuda_stream)
t_issue = time.perf_counter() - t0
side.synchronize()
dt = time.perf_counter() - t0
print(f"ctypes hipMemcpyAsync rc={rc}: 4 MiB x 50: {size * 50 / dt / 1e9:.1f} GB/s, issue {t_issue / 50 * 1e6:.1f} us/copy")
print("ulimit -l:", os.popen("ulimit -l").read().strip())
rc = torch.cuda.cudart().cudaHostUnregister(host.data_ptr())
print("unregister rc", rc)

"""Host-to-device staging options for host-resident signals: pageable copy, register-in-place, threaded copy to pinned."""
import time
import ctypes as C
import numpy as np
import torch
from concurrent.futures import ThreadPoolExecutor

n = 200_000_000  # 1.6 GB of float64
src = np.random.default_rng(0).standard_normal(n)
dst = torch.empty(n, dtype=torch.float64, device='cuda')
torch.cuda.synchronize()
t = torch.from_numpy(src)
for _ in range(2):
    t0 = time.perf_counter(); dst.copy_(t); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'pageable copy_          : {dt*1e3:7.1f} ms  {n*8/dt/1e9:6.1f} GB/s')
rt = torch.cuda.cudart()
t0 = time.perf_counter(); rc = rt.cudaHostRegister(src.ctypes.data, n * 8, 0); dt = time.perf_counter() - t0
print(f'hipHostRegister 1.6 GB  : {dt*1e3:7.1f} ms rc={rc}')
for _ in range(2):
    t0 = time.perf_counter(); dst.copy_(t, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'registered copy_        : {dt*1e3:7.1f} ms  {n*8/dt/1e9:6.1f} GB/s')
t0 = time.perf_counter(); rt.cudaHostUnregister(src.ctypes.data); print(f'unregister              : {(time.perf_counter()-t0)*1e3:7.1f} ms')
pin = torch.empty(n, dtype=torch.float64).pin_memory()
pn = pin.numpy()
for th in (1, 2, 4, 8, 16):
    ex = ThreadPoolExecutor(th)
    parts = np.linspace(0, n, th * 4 + 1).astype(np.int64)
    def cp(i):
        np.copyto(pn[parts[i]:parts[i + 1]], src[parts[i]:parts[i + 1]])
    for _ in range(2):
        t0 = time.perf_counter(); list(ex.map(cp, range(th * 4))); dt = time.perf_counter() - t0
    print(f'memcpy to pinned x{th:2d}    : {dt*1e3:7.1f} ms  {n*8/dt/1e9:6.1f} GB/s')
for _ in range(2):
    t0 = time.perf_counter(); dst.copy_(pin, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'pinned copy_            : {dt*1e3:7.1f} ms  {n*8/dt/1e9:6.1f} GB/s')

"""What a fresh device allocation through torch's caching allocator and a page-lock of a 16 MB memory-backed file cost on the box
(0.02-30 ms for 0.5-1 GB depending on what the allocator holds; 0.2 ms for pages this process has touched).  Usage: exp_alloc_probe.py"""
import time, torch
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
for mb in (16, 128, 512, 512, 1024):
    t=time.perf_counter(); x=torch.empty(mb<<20, dtype=torch.uint8, device='cuda'); torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(f'fresh {mb} MB: {dt*1e3:.2f} ms')
    keep=x
    del x
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    for mb in (512, 512):
        t=time.perf_counter(); x=torch.empty(mb<<20, dtype=torch.uint8, device='cuda'); dt=time.perf_counter()-t
        print(f'other stream fresh {mb} MB: {dt*1e3:.2f} ms'); 
        y=x
import numpy as np, mmap, os
# host register cost
f=open('/dev/shm/probe_arena','w+b'); f.truncate(16<<20); mm=mmap.mmap(f.fileno(), 16<<20)
a=np.frombuffer(mm, np.uint8); a[::4096]=1
t=time.perf_counter(); r=torch.cuda.cudart().cudaHostRegister(a.ctypes.data, 16<<20, 0); dt=time.perf_counter()-t
print('hostRegister 16 MB', r, f'{dt*1e3:.2f} ms')
t=time.perf_counter(); torch.cuda.cudart().cudaHostUnregister(a.ctypes.data); print(f'unregister {1e3*(time.perf_counter()-t):.2f} ms')
os.unlink('/dev/shm/probe_arena')

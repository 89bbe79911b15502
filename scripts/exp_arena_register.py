"""What page-locking the reader arenas costs the parent (BatchQueue._arena_tensor): 48 memory-backed files of 16 MB, 10 MB of each
written by another process; mapping with and without MAP_POPULATE, hipHostRegister, the first upload, on one thread and on four."""
import mmap
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

N, SIZE, USED = 48, 16 << 20, 10 << 20
rt = torch.cuda.cudart()
torch.cuda.init()
dev = torch.empty(SIZE // 2, dtype=torch.int16, device='cuda')
torch.cuda.synchronize()


def make(tag):
    code = (f"import os\nfor k in range({N}):\n f=open('/dev/shm/exp_arena_{tag}_%d'%k,'wb'); f.truncate({SIZE}); f.seek(0); "
            f"f.write(b'\\1'*{USED}); f.close()")
    subprocess.run([sys.executable, '-c', code], check=True)
    return [f'/dev/shm/exp_arena_{tag}_{k}' for k in range(N)]


def drop(paths):
    for p in paths:
        os.unlink(p)


def one(path, populate, length=SIZE):
    t0 = time.perf_counter()
    with open(path, 'r+b') as fh:
        mm = mmap.mmap(fh.fileno(), length, flags=mmap.MAP_SHARED | (mmap.MAP_POPULATE if populate else 0))
    view = np.frombuffer(mm, dtype=np.int16)
    t1 = time.perf_counter()
    rc = int(rt.cudaHostRegister(view.ctypes.data, length, 0))
    t2 = time.perf_counter()
    return mm, view, t1 - t0, t2 - t1, rc


for tag, populate, length in (('a', True, SIZE), ('b', False, SIZE), ('c', False, USED), ('d', True, USED)):
    paths = make(tag)
    t_map = t_reg = 0.0
    keep = []
    t0 = time.perf_counter()
    for p in paths:
        mm, view, a, b, rc = one(p, populate, length)
        assert rc == 0
        t_map += a
        t_reg += b
        keep.append((mm, view))
    wall = time.perf_counter() - t0
    t1 = time.perf_counter()
    for mm, view in keep:
        dev[:USED // 2].copy_(torch.from_numpy(view)[:USED // 2], non_blocking=True)
    torch.cuda.synchronize()
    up = time.perf_counter() - t1
    print(f'populate={populate!s:5} mapped {length >> 20} MB of 16: map {t_map * 1e3:6.1f} ms  register {t_reg * 1e3:6.1f} ms  wall {wall * 1e3:6.1f} ms  '
          f'upload of {N} x 10 MB {up * 1e3:6.1f} ms ({N * USED / up / 1e9:.1f} GB/s)', flush=True)
    for mm, view in keep:
        rt.cudaHostUnregister(view.ctypes.data)
    drop(paths)

paths = make('e')
keep = []
t0 = time.perf_counter()


def work(part):
    for p in part:
        keep.append(one(p, True))


ts = [threading.Thread(target=work, args=(paths[k::4],)) for k in range(4)]
for t in ts:
    t.start()
for t in ts:
    t.join()
print(f'four threads, populate, 16 MB: wall {(time.perf_counter() - t0) * 1e3:6.1f} ms', flush=True)
for mm, view, *_ in keep:
    rt.cudaHostUnregister(view.ctypes.data)
drop(paths)

# pageable upload for comparison (no registration at all)
paths = make('f')
maps = []
for p in paths:
    with open(p, 'r+b') as fh:
        maps.append(np.frombuffer(mmap.mmap(fh.fileno(), SIZE), dtype=np.int16))
t0 = time.perf_counter()
for v in maps:
    dev[:USED // 2].copy_(torch.from_numpy(v)[:USED // 2], non_blocking=True)
torch.cuda.synchronize()
up = time.perf_counter() - t0
print(f'pageable upload of {N} x 10 MB: {up * 1e3:6.1f} ms ({N * USED / up / 1e9:.1f} GB/s)', flush=True)
drop(paths)

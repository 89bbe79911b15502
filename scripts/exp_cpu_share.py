import os, time, subprocess, sys
print('cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else None)
for p in ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us','/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    if os.path.exists(p): print(p, open(p).read().strip())
print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())
code = "import time\nt=time.perf_counter()\nx=0\nfor i in range(6000000): x+=i*i\nprint(time.perf_counter()-t)"
for n in (1, 8, 16, 32, 64, 128):
    t=time.perf_counter()
    ps=[subprocess.Popen([sys.executable,'-c',code],stdout=subprocess.PIPE) for _ in range(n)]
    ts=[float(p.communicate()[0]) for p in ps]
    print(n, 'wall', round(time.perf_counter()-t,3), 'mean inner', round(sum(ts)/n,3), 'max', round(max(ts),3))

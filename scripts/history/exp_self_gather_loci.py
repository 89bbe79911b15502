import os, sys, tempfile, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from warpstr_amd.wrapper import main_wrapper_loci
root = tempfile.mkdtemp()
specs = [(f'l{i:04d}', bench.MANY_LOCI_PATTERNS[i % 10], 110, (2271, 3701), 5000 + i) for i in range(600)]
a, raws = bench.make_locus_dirs(os.path.join(root, 'a'), specs, 30, 77)
b, _ = bench.make_locus_dirs(os.path.join(root, 'b'), specs, 30, 77)
reader = lambda p: raws[os.path.basename(p)[:-6]]
t = time.perf_counter(); main_wrapper_loci(a, 8, raw_reader=reader, quiet=True); print('plain', time.perf_counter() - t)
tm = {}
t = time.perf_counter(); main_wrapper_loci(b, 8, raw_reader=reader, quiet=True, shard=True, timings=tm); print('one-rank RCCL group', time.perf_counter() - t, 'gather_s', tm['gather_s'])
import filecmp
bad = [x.name for x, y in zip(a, b) for rel in ('overview.csv', 'predictions/sequences/all.fasta') if not filecmp.cmp(os.path.join(x.path, rel), os.path.join(y.path, rel), shallow=False)]
print('differing loci:', bad[:5], len(bad))
import torch.distributed as d
d.destroy_process_group()

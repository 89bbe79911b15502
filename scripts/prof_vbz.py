"""wsx_vbz_decode alone on the upstream file's real chunks (bench.py's vbz_kernel_leg): HIP-event time per launch.  Usage: prof_vbz.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

r = bench.vbz_kernel_leg(os.path.join(bench.ROOT, 'tests', 'golden', 'real', 'batch_0.fast5'), 0)
print(json.dumps({k: r[k] for k in ('kernel', 'blocks_per_launch', 'launch_ms', 'roofline') if k in r}))

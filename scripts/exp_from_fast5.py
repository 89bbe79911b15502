"""The from_fast5 leg of bench.py by itself (N copies of the upstream test fast5 through main_wrapper_loci, files -> output files).
Usage: [WSX_SHARED_MB=.. WSX_SHARED_READS=..] exp_from_fast5.py [n_copies]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from warpstr_amd import loci

if os.environ.get('WSX_SHARED_MB'):   # experiment: raw bytes / reads of a batch the reader processes decode into a staging buffer
    loci.SHARED_BATCH_BYTES = int(os.environ['WSX_SHARED_MB']) << 20
    loci.SHARED_BATCH_READS = int(os.environ.get('WSX_SHARED_READS', loci.SHARED_BATCH_READS))

if os.environ.get('WSX_PROFILE'):   # experiment: where the calling thread of the 16-reader legs spends its time (cProfile, to stderr)
    import cProfile
    import io
    import pstats
    from warpstr_amd import wrapper
    _orig = wrapper.main_wrapper_loci

    def _profiled(loci_, threads, **kw):
        if threads <= 1 or len(loci_) < 64:
            return _orig(loci_, threads, **kw)
        pr = cProfile.Profile()
        pr.enable()
        try:
            return _orig(loci_, threads, **kw)
        finally:
            pr.disable()
            out = io.StringIO()
            pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(28)
            print(out.getvalue(), file=sys.stderr, flush=True)
    wrapper.main_wrapper_loci = _profiled
if os.environ.get('WSX_NO_GC'):   # experiment: are the phases that sometimes take 0.4 s longer full garbage collections?
    import gc
    gc.disable()
if os.environ.get('WSX_GC_DEBUG'):
    import gc
    import time
    _t = [0.0]

    def _cb(phase, info):
        if phase == 'start':
            _t[0] = time.perf_counter()
        elif info['generation'] == 2:
            print(f'gen-2 collection: {(time.perf_counter() - _t[0]) * 1e3:.0f} ms', file=sys.stderr, flush=True)
    gc.callbacks.append(_cb)
print(json.dumps(bench.from_fast5_leg(int(sys.argv[1]) if len(sys.argv) > 1 else 1500, 0), indent=1))

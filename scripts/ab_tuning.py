"""Same-box, same-process A/B of a launch-policy knob (wsx_caller_set_tuning) on bench.py's three workloads: K timed pipelined
steps per setting, settings alternating, the records of the last step of every setting compared byte for byte.
Usage: ab_tuning.py KNOB VALUE_A VALUE_B [steps] [rounds]      e.g.  ab_tuning.py mid_fork 0 1"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import bench  # noqa: E402


def main():
    import torch

    from warpstr_amd import _lib
    from warpstr_amd.caller import HipCaller
    knob, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    device = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    pat, fl, tr = bench.CFG1
    makers = (('headline', lambda: bench.make_headline(100000, 2000, 1000, device)),
              ('cfg1', lambda: bench.make_ragged('cfg1', [(pat, fl, tr, 1, None)], 20000, 1000, device)),
              ('cfg5', lambda: bench.make_ragged('cfg5', [(p, bench.cfg5_flank(p, 11 + i), (500, 5000), 11 + i, None)
                                                           for i, p in enumerate(bench.CFG5_PATTERNS)], 50000, 1000, device)))
    out = {'knob': knob, 'values': [va, vb], 'steps': steps, 'rounds': rounds}
    for name, make in makers:
        wl = make()
        hip = HipCaller(wl.tables, wl.flanks, device=0, stream=torch.cuda.current_stream().cuda_stream)
        hip.set_pipelined(True)
        res = [torch.zeros((wl.n, _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=device) for _ in range(bench.N_BUF)]
        ms = {va: [], vb: []}
        last = {}
        for r in range(rounds):
            for v in (va, vb) if r % 2 == 0 else (vb, va):
                hip.set_tuning(knob, v)
                for k in range(4):
                    hip.call_device(wl.signal.data_ptr(), wl.offsets, wl.aut, res[k % bench.N_BUF].data_ptr())
                hip.synchronize()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(steps):
                    hip.call_device(wl.signal.data_ptr(), wl.offsets, wl.aut, res[k % bench.N_BUF].data_ptr())
                hip.synchronize()
                torch.cuda.synchronize()
                ms[v].append((time.perf_counter() - t0) / steps * 1e3)
                last[v] = res[(steps - 1) % bench.N_BUF].cpu().numpy().tobytes()
        out[name] = {f'{knob}={v}': {'ms_per_step': [round(x, 3) for x in ms[v]], 'best': round(min(ms[v]), 3), 'median': round(float(np.median(ms[v])), 3)}
                     for v in (va, vb)}
        out[name]['records_identical'] = bool(last[va] == last[vb])
        hip.close()
        del wl
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""bench.py -- reads/s of the HIP caller on BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--samples T]

Workload (config.workload): BASELINE.json configs[2] -- 100k synthetic reads, 2 kSample squiggles, HD-style
interrupted automaton `(AGC)AACAGCCGCCAC(CGC)` with <= 64 states -- per GPU (weak scaling: reads shard with
no exchange on the data path; one RCCL all-gather collects the per-read result records each step).
A "step" is one full call of the batch: both DTW passes, rescaling fit, bad-repeat masking, allele lengths.
Inputs are resident in HBM before the timed region.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The HIP runtime multiplexes a process's streams onto 4 hardware queues by default.  The batch call uses five streams,
# the result gather a sixth; a stream that shares a queue with the gather's wait-for-step-k barrier cannot start its
# step k+1 work behind it (measured: +0.8 ms per step on the collective path).  Has to be set before HIP initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

PATTERN = '(AGC)AACAGCCGCCAC(CGC)'
FLANK = 19
HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VALU_PEAK = 78.6e12 / 2  # fp64 vector adds/s: 78.6 TFLOP/s counts FMA as 2


def make_workload(n_reads, T, seed, device):
    """Clean level sequences on the host (seeded), noise added on the device (seeded)."""
    import torch

    from warpstr_amd import synth
    locus = synth.make_locus(PATTERN, FLANK, 2024, max_states=64)
    rng = np.random.default_rng(seed)
    n_tpl = min(n_reads, 2048)
    pm_sigs, revs = [], []
    for _ in range(n_tpl):
        rev = bool(rng.random() < 0.5)
        s, _ = synth.squiggle(locus, rev, T, rng, sigma=0.0)
        pm_sigs.append(s)
        revs.append(rev)
    clean = torch.from_numpy(np.stack(pm_sigs)).to(device)
    idx = torch.from_numpy(rng.integers(0, n_tpl, size=n_reads)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    signal = clean[idx] + 0.25 * torch.randn((n_reads, T), generator=g, device=device, dtype=torch.float64)
    signal = signal.reshape(-1).contiguous()
    aut = np.array(revs, dtype=np.int32)[idx.cpu().numpy()]
    offsets = np.arange(n_reads + 1, dtype=np.int64) * T
    return locus, signal, offsets, aut


def cpu_baseline(locus, signal_host, T, aut, budget_s=15.0):
    """The CPU oracle (a C port of the reference algorithm; the Python reference cannot travel) on this
    box's host cores, on a bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle
    cores = os.cpu_count() or 1
    oa = [oracle.Automaton.from_table(locus.template, FLANK), oracle.Automaton.from_table(locus.reverse, FLANK)]
    oracle.lib()

    def one(i):
        return oracle.call_read(oa[aut[i]], signal_host[i * T:(i + 1) * T], debug=False).len2

    t0 = time.perf_counter()
    one(0)
    t1 = time.perf_counter() - t0
    n = int(max(cores, min(len(aut), budget_s * cores / max(t1, 1e-4))))
    n = min(n, len(aut))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL during the call
        list(ex.map(one, range(n)))
    dt = time.perf_counter() - t0
    return {'value': n / dt, 'unit': 'reads/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} reads of the same workload (T={T}), C oracle, {cores} threads'}


VALU_INSTS_PER_ROW = 10.51   # SQ_INSTS_VALU per DP row per wave (profiles/r01s5_pmc.json); the formulation's floor is 10
LDS_CYCLES_PER_ROW = 10.0    # 2 ds_read_b64 (2 cycles each) + 1 ds_write_b64 (~6): MI355X_MICROARCH.md, LDS table
N_SIMD, N_CU, CLK_MAX_HZ, CLK_OBSERVED_HZ = 1024, 256, 2.4e9, 1.89e9


def valu_roofline(tm1, n, T):
    """What actually bounds the fill: wave-level VALU instruction issue (every VALU op, fp64 or 32-bit, occupies a
    SIMD for 4 cycles per wave64).  Measured on an extra single-stream step (HIP events around the fill launches)."""
    launches = max(tm1['dp_launches'], 1)
    ms = tm1['dp_kernel_ms'] / launches
    rows = 2.0 * n * T / launches                      # wave-rows per launch (one wave per read)
    achieved = rows * VALU_INSTS_PER_ROW / (ms * 1e-3)  # wave-instructions per second
    peak = N_SIMD * CLK_MAX_HZ / 4.0
    lds = rows * LDS_CYCLES_PER_ROW / (ms * 1e-3)       # LDS-pipe cycles per second, all CUs
    return {'bound': 'valu-issue', 'achieved': achieved, 'peak': peak, 'unit': 'wave64 VALU instr/s',
            'frac': achieved / peak, 'frac_at_observed_clock': achieved / (N_SIMD * CLK_OBSERVED_HZ / 4.0),
            'launch_ms_alone': ms, 'launches': launches,
            'lds_pipe': {'cycles_per_row': LDS_CYCLES_PER_ROW, 'frac': lds / (N_CU * CLK_MAX_HZ),
                         'frac_at_observed_clock': lds / (N_CU * CLK_OBSERVED_HZ)},
            'note': 'peak = 1024 SIMDs x 2.4 GHz / 4 cycles; the chip holds ~1.89 GHz under this fp64 load '
                    '(GRBM_GUI_ACTIVE); the LDS pipe (one per CU: predecessor exchange) is loaded as heavily as the VALU'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--reads', type=int, default=100000)
    ap.add_argument('--samples', type=int, default=2000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    # Exactly one line on stdout: native libraries print there too (RCCL's start-up banner: version, hostname, library
    # path), so file descriptor 1 points at stderr until the JSON line is written to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU path)')
    # one process per GPU; WARPSTR_BENCH_BACKEND=gloo lets the multi-rank logic be exercised on a 1-GPU box
    backend = os.environ.get('WARPSTR_BENCH_BACKEND', 'nccl')
    local = local % torch.cuda.device_count() if backend != 'nccl' else local
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    # WARPSTR_BENCH_SELF_GATHER=1: a one-rank RCCL group on a 1-GPU box, to exercise the collective path of N > 1
    self_gather = world == 1 and bool(os.environ.get('WARPSTR_BENCH_SELF_GATHER'))
    if world > 1:
        if backend == 'nccl':
            # no device_id: binding the group to the device at init (eager communicator) cost every later step 2.3 ms on
            # this stack (measured with a one-rank group: 19.4 vs 17.3 ms per step); the communicator is created by the
            # first collective of the warm-up instead, on the device set above
            dist.init_process_group('nccl')
        else:
            dist.init_process_group(backend)
    elif self_gather:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29517', rank=0, world_size=1)
    collective = (world > 1 and backend == 'nccl') or self_gather

    from warpstr_amd import _lib
    from warpstr_amd.caller import HipCaller
    from warpstr_amd.dist import gather_results

    n, T = args.reads, args.samples
    locus, signal, offsets, aut = make_workload(n, T, 1000 + rank, device)
    stream = torch.cuda.current_stream().cuda_stream
    hip = HipCaller([locus.template, locus.reverse], [FLANK, FLANK], device=local, stream=stream,
                    workspace_limit=96 << 30)
    # Two result buffers: the all-gather of step k runs on a side stream while the kernels of step k+1 already fill the
    # other buffer (the collective moves 56 B per read and rank over xGMI: latency-bound, nothing for the CUs to do).
    res_bufs = [torch.zeros((n, _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=device) for _ in range(2)]
    gather_stream = torch.cuda.Stream(device=device) if collective else None
    gather_done = [None, None]
    step_no = [0]
    # back-to-back steps: a call no longer drains into the handle's stream, the next step's chunks follow on every
    # internal stream (wsx_caller_set_pipelined); consumers are ordered after a step with join()
    hip.set_pipelined(True)

    def step():
        k = step_no[0] & 1
        step_no[0] += 1
        results = res_bufs[k]
        if gather_done[k] is not None:  # the gather that read this buffer two steps ago
            torch.cuda.current_stream().wait_event(gather_done[k])
        hip.call_device(signal.data_ptr(), offsets, aut, results.data_ptr())
        if collective:
            hip.join(gather_stream.cuda_stream)  # every kernel of this step is ahead of the gather
            with torch.cuda.stream(gather_stream):
                out = torch.empty((world * n, results.shape[1]), dtype=torch.uint8, device=device)
                dist.all_gather_into_tensor(out, results)
                done = torch.cuda.Event()
                done.record()
            gather_done[k] = done
            return out
        if world > 1:  # test path (gloo): CPU collective
            hip.synchronize()
            return gather_results(results.cpu(), world)
        return results

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    hip.timing_window(True)   # the fill kernels' HIP events of ALL timed steps are kept (roofline.launch_ms)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        allres = step()
    hip.synchronize()        # pipelined calls end on the library's own streams
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = hip.last_timing()  # HIP events on the launch streams, every fill launch of the timed region
    hip.timing_window(False)
    # untimed extra step on ONE stream: the fill kernel's duration when nothing runs beside it (VALU roofline)
    hip.set_streams(1)
    step()
    torch.cuda.synchronize()
    tm1 = hip.last_timing()
    hip.set_streams(4)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        res = allres.cpu().numpy().view(_lib.RESULT_DTYPE).reshape(-1)
        ok = int((res['status'] == 0).sum())
        S = max(locus.template.n_states, locus.reverse.n_states)
        total_reads = n * world * args.steps
        reads_per_s = total_reads / dt
        # dominant kernel = the DTW fill.  One step = 2 passes over n reads, issued as `launches` kernel launches
        # (the library splits big batches into chunks that overlap on two streams); durations are HIP events
        # recorded on the launch streams around each fill launch.
        launches_total = max(tm['dp_launches'], 1)
        launches = launches_total / args.steps            # per step
        launch_ms = tm['dp_kernel_ms'] / launches_total   # average over every fill launch of the timed region
        reads_per_launch = 2.0 * n / launches
        algo_bytes_per_launch = reads_per_launch * (12 * T + 32) / 2.0   # 12T+32 B/read for both passes
        achieved = algo_bytes_per_launch / (launch_ms * 1e-3) / 1e9
        cells_per_s = reads_per_launch * T * S / (launch_ms * 1e-3)
        traffic = None  # HBM bytes per launch from rocprofv3 PMC passes (profiles/r01s5_traffic.json), same workload only
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01s5_traffic.json')) as f:
                tj = json.load(f)
            if tj['workload']['samples'] == T:
                traffic = tj['hbm_bytes_per_launch'] / tj['workload']['reads'] * reads_per_launch
        except (OSError, KeyError, ValueError):
            pass
        out = {
            'metric': 'reads/s (STR segments aligned)', 'value': reads_per_s, 'unit': 'reads/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[2]: {n} reads/GPU x {T} samples, {PATTERN} flank {FLANK}, '
                                   f'S={locus.template.n_states}/{locus.reverse.n_states} states, both passes',
                       'reads_per_gpu': n, 'samples_per_read': T, 'states': S, 'called_ok': ok,
                       'results_gather': ((f'{backend} all_gather of 56-B records per step' +
                                          (', overlapped with the next step' if collective else ' (CPU test path, synchronous)')) if world > 1
                                          else ('one-rank nccl group (self test)' if self_gather else 'none (1 GPU)'))},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'kernel': hip.kernel_name(0), 'launch_ms': launch_ms, 'launches_per_step': launches,
                         'reads_per_launch': reads_per_launch,
                         'note': 'min-plus recurrence: bound by fp64 VALU issue, not HBM (see valu)'},
            'valu_roofline': valu_roofline(tm1, n, T),
            'valu': {'dp_cells_per_s': cells_per_s, 'valu_insts_per_row_per_wave': VALU_INSTS_PER_ROW,
                     'note': 'PMC: SQ_INSTS_VALU = 10.51 per row per wave (floor of this formulation: 10 = 6 adds, 2 '
                             'compares, 2 mins); the fill launches overlap other chunks\' kernels on 4 streams, so '
                             'launch_ms is a co-scheduled duration (5.3-5.4 ms per 100k reads when the kernel runs alone)'},
            'dp_kernel_ms_per_step': tm['dp_kernel_ms'] / args.steps,
            # first enqueue to last finish of the timed region on the device clock (HIP events), per step
            'device_ms_per_step': tm['total_ms'] / args.steps,
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = signal[: min(n, 4096) * T].cpu().numpy()
            out['cpu_baseline'] = cpu_baseline(locus, sample, T, aut[: min(n, 4096)])
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if world > 1 or self_gather:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""bench.py -- reads/s of the HIP caller on BASELINE.json's workloads.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--workload headline|cfg1|cfg5]
                    [--reads R] [--samples T]

Workloads (config.workload)
  headline  BASELINE.json configs[2]: 100k synthetic reads, 2 kSample squiggles, HD-style interrupted automaton
            `(AGC)AACAGCCGCCAC(CGC)` with <= 64 states.  --scaling weak (default): that many reads PER GPU;
            --scaling strong = configs[3]: the SAME 100k-read workload sharded over the N GPUs
            (warpstr_amd.dist.shard_reads), one RCCL all-gather of the result records per step.
  cfg1      the shape of the upstream test case / every flank-110 locus: `(AAAT)` flank 110, S = 225 states,
            T in [2271, 3701] samples, 20k reads per GPU (kernel dtw_fill_fast<4, 4, 2, 1, false, 1>: lane-major).
  cfg5      BASELINE.json configs[4] at one GPU's share: 8 loci x 2 strands, ~128-state automata, T in [500, 5000],
            50k reads per GPU.
A "step" is one full call of the batch: both DTW passes, rescaling fit, bad-repeat masking, allele lengths.
Inputs are resident in HBM before the timed region.  One JSON line is printed by rank 0.  The results of the timed
steps are compared with the CPU oracle on a sample of the reads ("verified"); a mismatch exits non-zero.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The HIP runtime multiplexes a process's streams onto 4 hardware queues by default.  The batch call uses up to ten streams,
# the result gather another; a stream that shares a queue with the gather's wait-for-step-k barrier cannot start its
# step k+1 work behind it (measured: +0.8 ms per step on the collective path).  Has to be set before HIP initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
N_SIMD, N_CU, CLK_MAX_HZ = 1024, 256, 2.4e9
# Measured issue cost of the fill's instruction mix (6 v_add_f64, 2 v_cmp_lt_f64 -> SGPR pair, 2 v_min_f64, dependent as in the
# kernel) with 8 waves per SIMD and nothing else to do: 4.29 SIMD cycles per wave-instruction (scripts/exp_valu_rate.hip under
# rocprofv3 --pmc, profiles/r02_valu_rate.log; plain v_add/v_min 4.15, the compare into an SGPR pair 4.51) -- a property of
# the hardware, not of the kernel: the nominal 4 cycles are not reachable for this mix.
ROW_MIX_CYCLES_PER_INST = 4.29
N_BUF = 4  # result buffers >= pipelined calls in flight (two for big batches, four for small ones: include/warpstr_hip.h)

HEADLINE = ('(AGC)AACAGCCGCCAC(CGC)', 19)
CFG1 = ('(AAAT)', 110, (2271, 3701))
CFG5_PATTERNS = ['((CAGG){CAGM})(CAGA)(CA)', '(CAG)CAACAG(CCG)', '(GGCCCC)', '(CTG)CTA(CTG)', '(AAGGG)(AAAGG)', '(CCTG)(TCTG)',
                 '(GAA)', '(CAG)(CAA)(CAG)']


class Workload:
    """Automata + device-resident reads of one rank."""

    def __init__(self, name, desc, tables, flanks, signal, offsets, aut, oracle_automata):
        self.name, self.desc = name, desc
        self.tables, self.flanks = tables, flanks
        self.signal, self.offsets, self.aut = signal, offsets, aut
        self.oracle_automata = oracle_automata  # () -> list of oracle.Automaton (built lazily: the oracle is the checker)

    @property
    def n(self):
        return len(self.aut)


def _noisy(clean_rows, lens, idx, seed, device):
    """signal = clean template read + N(0, 0.25) noise, on the device (seeded): ragged rows -> one flat f64 tensor."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    flat = torch.from_numpy(np.concatenate([clean_rows[i] for i in idx])).to(device)
    flat += 0.25 * torch.randn(flat.shape, generator=g, device=device, dtype=torch.float64)
    offsets = np.zeros(len(idx) + 1, np.int64)
    np.cumsum(lens[idx], out=offsets[1:])
    return flat.contiguous(), offsets


def make_headline(n_local, T, seed, device, picks=None):
    """configs[2].  `picks` (strong scaling): template index of each of this rank's reads, drawn from the global
    workload; else n_local reads drawn with this rank's seed."""
    import torch

    from warpstr_amd import synth
    pattern, fl = HEADLINE
    locus = synth.make_locus(pattern, fl, 2024, max_states=64)
    rng = np.random.default_rng(1000)  # templates are the same on every rank
    n_tpl = 2048
    pm_sigs, revs = [], []
    for _ in range(n_tpl):
        rev = bool(rng.random() < 0.5)
        s, _ = synth.squiggle(locus, rev, T, rng, sigma=0.0)
        pm_sigs.append(s)
        revs.append(rev)
    if picks is None:
        picks = np.random.default_rng(seed).integers(0, n_tpl, size=n_local)
    clean = torch.from_numpy(np.stack(pm_sigs)).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    idx = torch.from_numpy(picks).to(device)
    signal = clean[idx] + 0.25 * torch.randn((len(picks), T), generator=g, device=device, dtype=torch.float64)
    signal = signal.reshape(-1).contiguous()
    aut = np.array(revs, dtype=np.int32)[picks]
    offsets = np.arange(len(picks) + 1, dtype=np.int64) * T

    def oa():
        from oracle import oracle
        return [oracle.Automaton.from_table(locus.template, fl), oracle.Automaton.from_table(locus.reverse, fl)]
    desc = (f'{pattern} flank {fl}, S={locus.template.n_states}/{locus.reverse.n_states} states, {T} samples per read')
    return Workload('headline', desc, [locus.template, locus.reverse], [fl, fl], signal, offsets, aut, oa)


def make_ragged(name, loci_spec, n_local, seed, device):
    """cfg1 / cfg5: loci_spec = [(pattern, flank, (Tmin, Tmax), locus seed, max_states or None)]; reads are dealt evenly
    over the loci and strands; 96 clean template reads per locus, noise added on the device."""
    from warpstr_amd import synth
    rng = np.random.default_rng(seed)
    tables, flanks, clean, clean_aut = [], [], [], []
    for li, (pattern, fl, (tmin, tmax), lseed, max_states) in enumerate(loci_spec):
        locus = synth.make_locus(pattern, fl, lseed, max_states=max_states)
        tables += [locus.template, locus.reverse]
        flanks += [fl, fl]
        trng = np.random.default_rng(7000 + li)
        for _ in range(96):
            rev = bool(trng.random() < 0.5)
            t = int(trng.integers(tmin, tmax + 1))
            hi = max(1, min(30, (t // 4 - 2 * fl - 12) // 14))
            clean.append(synth.squiggle(locus, rev, t, trng, lo=1, hi=hi, sigma=0.0)[0])
            clean_aut.append(2 * li + int(rev))
    lens = np.array([len(c) for c in clean], np.int64)
    idx = rng.integers(0, len(clean), size=n_local)
    signal, offsets = _noisy(clean, lens, idx, seed, device)
    aut = np.array(clean_aut, np.int32)[idx]

    def oa():
        from oracle import oracle
        return [oracle.Automaton.from_table(t, f) for t, f in zip(tables, flanks)]
    S = [t.n_states for t in tables]
    desc = (f'{len(loci_spec)} loci x 2 strands ({", ".join(p for p, *_ in loci_spec[:3])}{", ..." if len(loci_spec) > 3 else ""}), '
            f'S={min(S)}..{max(S)} states, T in [{loci_spec[0][2][0]}, {loci_spec[0][2][1]}] samples')
    return Workload(name, desc, tables, flanks, signal, offsets, aut, oa)


def cfg5_flank(pattern, seed):
    """configs[4] asks for 128-state automata: the longest flank with which both strands' automata have <= 128 states."""
    from warpstr_amd import synth
    for fl in range(64, 20, -1):
        locus = synth.make_locus(pattern, fl, seed)
        if max(locus.template.n_states, locus.reverse.n_states) <= 128:
            return fl
    raise RuntimeError(f'no flank length gives <= 128 states for {pattern}')


def oracle_sample(wl, signal_host, n_max, budget_s):
    """The CPU oracle (a C port of the reference algorithm; the Python reference cannot travel) on this box's host
    cores, on the first reads of the workload: returns (results per read, cpu_baseline record)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle
    cores = os.cpu_count() or 1
    oa = wl.oracle_automata()
    oracle.lib()
    off, aut = wl.offsets, wl.aut

    def one(i):
        return oracle.call_read(oa[aut[i]], signal_host[off[i]:off[i + 1]], debug=False)

    t0 = time.perf_counter()
    first = one(0)
    t1 = time.perf_counter() - t0
    n = int(max(min(cores, n_max), min(n_max, budget_s * cores / max(t1, 1e-4))))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL during the call
        res = list(ex.map(one, range(n)))
    dt = time.perf_counter() - t0
    res[0] = first
    base = {'value': n / dt, 'unit': 'reads/s', 'cores': cores, 'kind': 'port',
            'sample': f'the first {n} reads of the same workload ({wl.name}), C oracle, {cores} threads'}
    return res, base


def verify(gpu_records, oracle_results):
    """Result records of the timed run against the oracle, read by read: status and both allele lengths identical, both
    state-wise costs within 1e-5 relative (north_star's tolerance)."""
    bad = []
    for i, o in enumerate(oracle_results):
        g = gpu_records[i]
        ok = int(g['status']) == o.status
        if ok and o.status == 0:
            ok = (int(g['len1']), int(g['len2'])) == (o.len1, o.len2)
            for a, b in ((float(g['cost1']), o.cost1), (float(g['cost2']), o.cost2)):
                ok = ok and abs(a - b) <= 1e-5 * max(abs(b), 1e-300)
        if not ok:
            bad.append(i)
    return {'reads': len(oracle_results), 'mismatches': len(bad), 'first_mismatches': bad[:5],
            'fields': 'status, len1, len2 identical; cost1, cost2 within 1e-5 relative', 'against': 'oracle/ (CPU)'}


def union_ms(begin, end):
    """Total length of the union of intervals."""
    order = np.argsort(begin)
    tot, cs, ce = 0.0, None, None
    for i in order:
        b, e = float(begin[i]), float(end[i])
        if cs is None or b > ce:
            if cs is not None:
                tot += ce - cs
            cs, ce = b, e
        else:
            ce = max(ce, e)
    return tot + ((ce - cs) if cs is not None else 0.0)


def kernel_source_hash():
    """What the committed counters are valid for: the text of the fill kernels and of the state placement."""
    import hashlib
    h = hashlib.sha256()
    for rel in ('warpstr_amd/csrc/dtw_kernels.hip', 'warpstr_amd/csrc/wsx_place.h'):
        with open(os.path.join(ROOT, rel), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def fill_profile(kernel):
    """Counters of the fill kernel from the committed rocprofv3 PMC passes (profiles/fill_pmc.json, written by
    scripts/summarize_profiles.py).  The bench refuses to quote counters of a different kernel: an entry is keyed by the
    kernel's name AND carries the hash of the sources it was measured on (kernel_source_hash); an entry taken from other
    sources is not quoted (returns {'stale': ...}: the line then says so instead of carrying its numbers)."""
    path = os.path.join(ROOT, 'profiles', 'fill_pmc.json')
    with open(path) as f:
        table = json.load(f)
    entry = table.get(kernel)
    if entry is not None and entry.get('kernel_source_hash') != kernel_source_hash() and not os.environ.get('WARPSTR_BENCH_PROFILING'):
        return {'stale': f"profiles/fill_pmc.json: the counters of {kernel} were taken on other kernel sources "
                         f"(hash {entry.get('kernel_source_hash')}, now {kernel_source_hash()}); re-run scripts/profile_round.sh"}
    if kernel not in table and kernel.endswith(', 0>'):  # profiles taken before the kernel grew its last template parameter
        legacy = kernel[:-len(', 0>')] + '>'
        if legacy in table:
            return table[legacy]
    if kernel not in table:
        if os.environ.get('WARPSTR_BENCH_PROFILING'):  # scripts/profile_round.sh: the run that produces the entry
            return None
        raise SystemExit(f'bench.py: profiles/fill_pmc.json holds no PMC profile of {kernel} (has: {sorted(table)}); '
                         're-run scripts/profile_round.sh for this workload')
    return table[kernel]


def fill_traffic_ratio(kernel, mean_T):
    try:
        prof = fill_profile(kernel)
    except SystemExit:
        return None
    if not prof or prof.get('stale') or prof.get('hbm_bytes_per_sample') is None:
        return None
    return prof['hbm_bytes_per_sample'] / (6.0 + 16.0 / max(mean_T, 1.0))


def valu_roofline(prof, alone_ms, wave_rows, kernel):
    """What actually bounds the fill: wave-level VALU instruction issue (every VALU op, fp64 or 32-bit, occupies a
    SIMD for 4 cycles per wave64) together with the LDS pipe.  alone_ms: one fill launch with nothing beside it."""
    import re
    m = re.match(r'dtw_fill_(?:fast|wg)<(\d+), (\d+), (\d+), (\d+)(?:, (?:true|false), (\d+))?', kernel)
    K, F, FL = (int(m.group(2)), int(m.group(3)), int(m.group(4))) if m else (1, 2, 2)
    lm = int(m.group(5)) if m and m.group(5) else 0
    # ds_write_b64 ~6 cycles, ds_write2_b64 ~10, ds_read_b64 2 (MI355X_MICROARCH.md, LDS table).  Lane-major placement
    # (lm): only slot 0 reads LDS (F reads); slots 0 and K-1 (lm = 1: one ds_write2_b64), slot 1 as well (lm = 3) or all slots (lm = 2) write
    lds_cycles = (6.0 * K + 2.0 * (F + (K - 1) * FL)) if lm == 0 else ({1: 10.0, 3: 16.0}.get(lm, 6.0 * K) + 2.0 * F)
    vpr, clk = prof['valu_insts_per_wave_row'], prof['clock_hz_observed']
    achieved = wave_rows * vpr / (alone_ms * 1e-3)
    peak = N_SIMD * CLK_MAX_HZ / 4.0
    lds = wave_rows * lds_cycles / (alone_ms * 1e-3)
    return {'bound': 'valu-issue', 'achieved': achieved, 'peak': peak, 'unit': 'wave64 VALU instr/s',
            'frac': achieved / peak, 'frac_at_observed_clock': achieved / (N_SIMD * clk / 4.0),
            'frac_of_measured_issue_rate_at_observed_clock': achieved / (N_SIMD * clk / ROW_MIX_CYCLES_PER_INST),
            'launch_ms_alone': alone_ms, 'valu_insts_per_wave_row': vpr, 'clock_hz_observed': clk,
            'counters_from': prof['source'],
            'lds_pipe': {'cycles_per_row': lds_cycles, 'frac': lds / (N_CU * CLK_MAX_HZ),
                         'frac_at_observed_clock': lds / (N_CU * clk)},
            'note': 'peak = 1024 SIMDs x 2.4 GHz / 4 cycles; clock_hz_observed = GRBM_GUI_ACTIVE / launch time under this '
                    'fp64 load; the LDS pipe (one per CU: predecessor exchange) is loaded as heavily as the VALU'}


def from_raw_leg(hip, wl, device, steps, warmup):
    """north_star's "loads of raw-signal segments": the same reads as int16 DAC values -> wsx_prepare_signals (spike removal,
    MAD normalisation, slice; Fast5.get_data_processed, src/schemas/fast5.py:45-57) -> wsx_call_batch with the called sequences
    requested -> result records on the host.  Two variants: raw segments resident in HBM, and in pinned host memory (the
    upload of step k+1 runs on a copy stream beside the kernels of step k).  Checked against the float64 path: the same
    reads prepared by the host restatement of the loader and called through host buffers give identical records."""
    import torch

    from warpstr_amd import _lib
    from warpstr_amd.signal_prep import process_raw
    n, total = wl.n, int(wl.offsets[-1])
    raw_dev = torch.clamp(torch.round(wl.signal * 70.0 + 500.0), 0, 2047).to(torch.int16).contiguous()
    raw_host = raw_dev.cpu().pin_memory()
    lens = np.diff(wl.offsets)
    seg_lo, seg_hi = np.zeros(n, np.int64), lens - 1  # the whole uploaded segment (reads are cut to the STR region upstream)
    raw_bufs = [torch.empty_like(raw_dev) for _ in range(2)]
    sig_bufs = [torch.empty(total, dtype=torch.float64, device=device) for _ in range(2)]
    seq_bufs = [[torch.zeros(total, dtype=torch.uint8, device=device) for _ in range(2)] for _ in range(2)]
    res_bufs = [torch.zeros((n, _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=device) for _ in range(2)]
    res_host = [torch.empty((n, _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8).pin_memory() for _ in range(2)]
    copy_stream, down_stream = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
    main = torch.cuda.current_stream()

    def run(host_resident):
        uploaded, downloaded, prepared = [None, None], [None, None], [None, None]

        def upload(k):
            with torch.cuda.stream(copy_stream):
                if prepared[k & 1] is not None:
                    copy_stream.wait_event(prepared[k & 1])  # the loader of step k-2 has read this buffer
                raw_bufs[k & 1].copy_(raw_host, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            uploaded[k & 1] = ev

        def step(k):
            b = k & 1
            if downloaded[b] is not None:
                main.wait_event(downloaded[b])  # step k-2 used these signal / sequence / result buffers
            if host_resident:
                main.wait_event(uploaded[b])
            src = raw_bufs[b] if host_resident else raw_dev
            hip.prepare_device(src.data_ptr(), wl.offsets, seg_lo, seg_hi, sig_bufs[b].data_ptr(), wl.offsets)
            prepared[b] = torch.cuda.Event()
            prepared[b].record(main)
            if host_resident:
                upload(k + 1)  # beside this step's kernels
            hip.call_device(sig_bufs[b].data_ptr(), wl.offsets, wl.aut, res_bufs[b].data_ptr(), seq1_ptr=seq_bufs[b][0].data_ptr(),
                            seq2_ptr=seq_bufs[b][1].data_ptr())
            hip.join(down_stream.cuda_stream)
            with torch.cuda.stream(down_stream):
                res_host[b].copy_(res_bufs[b], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            downloaded[b] = ev

        if host_resident:
            upload(0)
        for k in range(warmup):
            step(k)
        hip.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(warmup, warmup + steps):
            step(k)
        hip.synchronize()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, (warmup + steps - 1) & 1

    dt_hbm, last = run(False)
    dt_host, last = run(True)
    got = res_host[last].numpy().view(_lib.RESULT_DTYPE).reshape(-1).copy()
    seq2 = seq_bufs[last][1].cpu().numpy()
    # the float64 path on the same reads: host restatement of the loader, host-buffer call
    nv = min(n, 512)
    raws = raw_host.numpy()
    f64 = np.concatenate([process_raw(raws[wl.offsets[i]:wl.offsets[i + 1]], (0, int(lens[i]) - 1), 'Brute') for i in range(nv)])
    want, extra = hip.call(f64, wl.offsets[:nv + 1], wl.aut[:nv], want_seqs=True)
    same = bool(got[:nv].tobytes() == want.tobytes()) and bool(np.array_equal(seq2[: len(f64)], extra['seq2']))
    return {'reads_per_s_hbm_int16': n * steps / dt_hbm, 'ms_per_step_hbm_int16': dt_hbm / steps * 1e3,
            'reads_per_s_host_int16': n * steps / dt_host, 'ms_per_step_host_int16': dt_host / steps * 1e3,
            'h2d_bytes_per_step': int(total) * 2, 'd2h_bytes_per_step': n * _lib.RESULT_DTYPE.itemsize,
            'steps': steps, 'identical_to_f64_path': {'reads': nv, 'identical': same},
            'pipeline': 'int16 segments -> wsx_prepare_signals (Brute spike removal, MAD normalisation) -> wsx_call_batch with '
                        'seq1/seq2 -> 56-B records to pinned host memory; host variant: upload of step k+1 on a copy stream'}


def secondary_leg(wl, local, device, steps, warmup, n_verify):
    """One more workload under the same clock rules (pipelined device-resident calls, K timed steps between two
    synchronisations), its records checked against the oracle: the production shape (cfg1) and configs[4]'s share (cfg5)
    ride along with the default run so that the driver's record holds them."""
    import torch

    from warpstr_amd import _lib
    from warpstr_amd.caller import HipCaller
    n = wl.n
    hip = HipCaller(wl.tables, wl.flanks, device=local, stream=torch.cuda.current_stream().cuda_stream)  # (the library's default limits)
    res = [torch.zeros((n, _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=device) for _ in range(N_BUF)]
    hip.set_pipelined(True)
    for k in range(warmup):
        hip.call_device(wl.signal.data_ptr(), wl.offsets, wl.aut, res[k % N_BUF].data_ptr())
    hip.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        hip.call_device(wl.signal.data_ptr(), wl.offsets, wl.aut, res[k % N_BUF].data_ptr())
    hip.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    got = res[(steps - 1) % N_BUF].cpu().numpy().view(_lib.RESULT_DTYPE).reshape(-1)
    nv = min(n, n_verify)
    ores, _ = oracle_sample(wl, wl.signal[: int(wl.offsets[nv])].cpu().numpy(), nv, 2.0)
    kernels = sorted({hip.kernel_name(a) for a in range(len(wl.tables))})
    out = {'workload': wl.desc, 'reads_per_step': n, 'steps': steps, 'warmup': warmup, 'ms_per_step': dt / steps * 1e3,
           'value': n * steps / dt, 'unit': 'reads/s', 'kernels': kernels,
           'called_ok': int((got['status'] == 0).sum()), 'verified': verify(got, ores),
           # HBM bytes of a fill launch (FETCH_SIZE + WRITE_SIZE, separate PMC passes) over its algorithmic bytes (6 T + 16 per
           # read and pass), per kernel, from the committed counters -- null where they were taken on other kernel sources
           'traffic_over_algorithmic': {k: fill_traffic_ratio(k, float(wl.offsets[-1]) / n) for k in kernels}}
    hip.close()
    return out


MANY_LOCI_PATTERNS = ['(AAAT)', '(AGC)AACAGCCGCCAC(CGC)', '((CAGG){CAGM})(CAGA)(CA)', '(AGC)', '(GGCCCC)', '(CCTG)(TCTG)',
                      '(AAGGG)(AAAGG)', '(GAA)', '(CAG)CAACAG(CCG)', '(CTG)CTA(CTG)']


def make_locus_dirs(root, specs, reads_per_locus, seed, device=None, info=None):
    """Locus directories as steps 1-2 of the pipeline leave them (overview.csv with the `saved` rows, the flank file) for
    specs = [(name, pattern, flank, (Tmin, Tmax), locus seed)], and the reads as raw int16 DAC segments in host memory
    ({read name: array}; l_start_raw = 0, r_end_raw = len - 1).  Six clean template reads per locus, noise per read.
    device: draw the noise on that GPU, a clean template's reads at a time (400 000 reads: seconds instead of a minute);
    info: a dict that receives per locus name (template index of every read, the synth locus)."""
    import pandas as pd

    from warpstr_amd import overview as ov, synth
    from warpstr_amd.wrapper import LocusPath
    rng = np.random.default_rng(seed)
    loci, raws = [], {}
    for name, pattern, fl, (tmin, tmax), lseed in specs:
        locus = synth.make_locus(pattern, fl, lseed)
        tpl = []
        for _ in range(6):
            rev = bool(rng.random() < 0.5)
            t = int(rng.integers(tmin, tmax + 1))
            hi = max(1, min(30, (t // 4 - 2 * fl - 12) // 14))
            tpl.append((rev, synth.squiggle(locus, rev, t, rng, lo=1, hi=hi, sigma=0.0)[0]))
        loc = os.path.join(root, name)
        ov.store_flanks(loc, [locus.left_t, locus.right_t, locus.left_r, locus.right_r])
        names = [f'{name}_r{i:05d}' for i in range(reads_per_locus)]
        pick = rng.integers(0, len(tpl), size=reads_per_locus)
        if info is not None:
            info[name] = (pick, locus)
        lens = []
        if device is not None:
            import torch
            g = torch.Generator(device=device)
            g.manual_seed(int(rng.integers(0, 2 ** 31)))
            lens = [len(tpl[k][1]) for k in pick]
            for k in range(len(tpl)):
                rows = np.flatnonzero(pick == k)
                if not len(rows):
                    continue
                clean = torch.from_numpy(tpl[k][1]).to(device)
                x = clean[None, :] + 0.25 * torch.randn((len(rows), len(clean)), generator=g, device=device, dtype=torch.float64)
                block = torch.clamp(torch.round(x * 70.0 + 500.0), 0, 2047).to(torch.int16).cpu().numpy()
                for q, i in enumerate(rows):
                    raws[names[i]] = block[q]
        for nm, k in (zip(names, pick) if device is None else ()):
            x = tpl[k][1] + 0.25 * rng.standard_normal(len(tpl[k][1]))
            raws[nm] = np.clip(np.round(x * 70.0 + 500.0), 0, 2047).astype(np.int16)
            lens.append(len(x))
        pd.DataFrame({'read_name': names, 'run_id': 0, 'reverse': [tpl[k][0] for k in pick], 'saved': 1, 'l_start_raw': 0,
                      'r_end_raw': [n - 1 for n in lens]}).to_csv(os.path.join(loc, 'overview.csv'), index=False)
        loci.append(LocusPath(loc, pattern, fl, name))
    return loci, raws


def scratch_dir():
    """Where the driver legs put their locus directories: memory-backed when the box has it (the legs time this package's host
    work, not the container's overlay file system: five small files per locus)."""
    return '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else None


def _driver_timings(tm, n_loci):
    per = lambda k: tm.get(k, 0.0) / max(n_loci, 1) * 1e3
    return {'wall_s': tm['total_s'], 'host_threads': tm.get('host_threads', 1), 'reader_processes': tm.get('reader_processes', 0),
            'native_overviews': tm.get('native_overviews'),
            # CPU time per locus (summed over the threads when there are several), and the wall-clock of the two phases
            'per_locus_ms': {'overview_csv': per('overview_s'), 'automaton_compile': per('automata_s'), 'state_similarity': per('similarity_s'),
                             'setup_wall': per('setup_wall_s'), 'outputs_wall': per('store_s')},
            'once_s': {'handle_create_placement': tm['handle_s'], 'inside_wsx_caller_create': tm.get('handle_create_s')},
            'batches_s': {'host_reads': tm['read_s'], 'submit': tm['submit_s'], 'wait_for_gpu': tm['collect_s']},
            'workspace_bytes': tm.get('workspace_bytes'), 'workspace_limit_bytes': tm.get('workspace_limit_bytes'),
            'kernels': tm.get('kernels')}


def many_loci_leg(n_loci, reads_per_locus, n_loop, local):
    """The regime real runs live in: hundreds to thousands of loci with tens of reads each (upstream: one main_wrapper call
    per locus, WarpSTR.py:33-76).  n_loci loci at the default flank 110, mixed patterns, reads_per_locus reads of 2271-3701
    samples each, from raw int16 reads in host memory to every locus's output files, through ONE handle
    (warpstr_amd.wrapper.main_wrapper_loci) -- and, for the first n_loop of them, through one handle per locus
    (main_wrapper in a loop); the files of those loci must be identical."""
    import filecmp
    import shutil
    import tempfile

    from warpstr_amd.wrapper import main_wrapper_loci
    root = tempfile.mkdtemp(prefix='wsx_many_loci_', dir=scratch_dir())
    try:
        specs = [(f'locus{i:04d}', MANY_LOCI_PATTERNS[i % len(MANY_LOCI_PATTERNS)], 110, (2271, 3701), 5000 + i) for i in range(n_loci)]
        t0 = time.perf_counter()
        loci, raws = make_locus_dirs(os.path.join(root, 'batched'), specs, reads_per_locus, 77)
        loop_loci, _ = make_locus_dirs(os.path.join(root, 'loop'), specs[:n_loop], reads_per_locus, 77)
        gen_s = time.perf_counter() - t0
        reader = lambda path: raws[os.path.basename(path)[:-len('.fast5')]]
        # warm-up on directories of its own (a second pass over a locus finds the first one's columns in its overview.csv):
        # first use of the kernels' code objects, pinned staging
        warm_loci, _ = make_locus_dirs(os.path.join(root, 'warm'), specs[:10], reads_per_locus, 77)
        main_wrapper_loci(warm_loci, 1, raw_reads=raws, device=local, quiet=True)
        # the run with ONE host thread on directories of its own, then the run whose files are compared, with 16
        single_loci, _ = make_locus_dirs(os.path.join(root, 'single'), specs, reads_per_locus, 77)
        tm1 = {}
        main_wrapper_loci(single_loci, 1, raw_reads=raws, device=local, quiet=True, timings=tm1)
        tm = {}
        workers = min(16, os.cpu_count() or 1)
        main_wrapper_loci(loci, workers, raw_reads=raws, device=local, quiet=True, timings=tm)
        n_reads = n_loci * reads_per_locus
        out = {'workload': f'{n_loci} loci x {reads_per_locus} reads, flank 110, {len(MANY_LOCI_PATTERNS)} patterns '
                           f'({", ".join(MANY_LOCI_PATTERNS[:3])}, ...), T in [2271, 3701], raw int16 reads in host memory -> output files (under ' + (scratch_dir() or 'the default temporary directory') + ')',
               'loci': n_loci, 'reads': n_reads, 'loci_per_s': n_loci / tm['total_s'], 'reads_per_s': n_reads / tm['total_s'],
               'one_handle': _driver_timings(tm, n_loci),
               'one_handle_one_thread': dict(_driver_timings(tm1, n_loci), loci_per_s=n_loci / tm1['total_s'], reads_per_s=n_reads / tm1['total_s']),
               'generation_s': gen_s}
        import contextlib
        import io
        # the per-locus loop, with main_wrapper's own steps timed one by one (src/caller/wrapper.py:17-41)
        from warpstr_amd import overview as ov
        from warpstr_amd.caller import CallerWrapper
        from warpstr_amd.wrapper import _store_outputs, get_raw_workload
        parts = {'overview_csv': 0.0, 'automata_handle_placement': 0.0, 'read_raw': 0.0, 'first_call_with_allocations': 0.0, 'outputs': 0.0,
                 'handle_teardown': 0.0}
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            for locus in loop_loci:
                t = [time.perf_counter()]
                overview_path, df_overview = ov.load_overview(locus.path)
                t.append(time.perf_counter())
                cw = CallerWrapper(locus, 1, device=local)
                t.append(time.perf_counter())
                names, reverses, raws_l, positions = get_raw_workload(df_overview, locus.path, reader)
                t.append(time.perf_counter())
                results = cw.run_raw(names, reverses, raws_l, positions, 'Brute')
                t.append(time.perf_counter())
                _store_outputs(locus, overview_path, df_overview, results, reverses, write=True)
                t.append(time.perf_counter())
                cw.hip.close()
                t.append(time.perf_counter())
                for k, (x, y) in zip(parts, zip(t[:-1], t[1:])):
                    parts[k] += y - x
        dt = time.perf_counter() - t0
        same = True
        for a, b in zip(loci[:n_loop], loop_loci):
            for rel in ('overview.csv', 'predictions/sequences/all.fasta', 'summaries/state_similarity.csv'):
                same = same and filecmp.cmp(os.path.join(a.path, rel), os.path.join(b.path, rel), shallow=False)
        out['one_handle_per_locus'] = {'loci': n_loop, 'ms_per_locus': dt / max(n_loop, 1) * 1e3, 'loci_per_s': n_loop / dt,
                                       'reads_per_s': n_loop * reads_per_locus / dt,
                                       'per_locus_ms': {k: v / max(n_loop, 1) * 1e3 for k, v in parts.items()},
                                       'note': 'main_wrapper per locus: automata, a handle with its streams, first-call allocations and '
                                               'a drained GPU per locus'}
        out['ms_per_locus'] = tm['total_s'] / n_loci * 1e3
        out['speedup_over_per_locus_handles'] = (dt / max(n_loop, 1)) / (tm['total_s'] / n_loci)
        out['outputs_identical'] = {'loci': n_loop, 'identical': bool(same)}
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def cfg5_driver_leg(reads_per_locus, local):
    """configs[4]'s share of one GPU (8 loci x 2 strands, ~128-state automata, T in [500, 5000]) through the PRODUCT seam:
    main_wrapper_loci from raw int16 reads in host memory to the eight loci's output files (the kernel-level figure with the
    reads resident in HBM is secondary.cfg5)."""
    import shutil
    import tempfile

    from warpstr_amd.wrapper import main_wrapper_loci
    root = tempfile.mkdtemp(prefix='wsx_cfg5_', dir=scratch_dir())
    try:
        specs = [(f'locus{i}', p, cfg5_flank(p, 11 + i), (500, 5000), 11 + i) for i, p in enumerate(CFG5_PATTERNS)]
        loci, raws = make_locus_dirs(root, specs, reads_per_locus, 78)
        main_wrapper_loci(loci, 1, raw_reads=raws, device=local, quiet=True)  # warm-up (code objects, pinned staging)
        tm = {}
        tables = main_wrapper_loci(loci, min(8, os.cpu_count() or 1), raw_reads=raws, device=local, quiet=True, timings=tm)
        n = len(loci) * reads_per_locus
        called = int(sum((np.asarray(df['results']) >= 0).sum() for df, _ in tables))
        return {'workload': f'8 loci x {reads_per_locus} reads through main_wrapper_loci, raw int16 reads in host memory -> output files',
                'reads': n, 'called_ok': called, 'reads_per_s': n / tm['total_s'], 'driver': _driver_timings(tm, len(loci))}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def cfg5_full_leg(reads_per_locus, local):
    """configs[4] at its FULL size on ONE GPU: 8 loci x 2 strands x reads_per_locus (50 000) reads = 400 000 reads of 500-5 000
    samples, ~128-state automata, through the product seam in one handle (main_wrapper_loci: raw int16 reads in host memory ->
    every locus's output files).  Run twice on directories of their own: the second run is the timed one and its files must equal
    the first's byte for byte (determinism).  Checked against the oracle on a sample of every locus (the loader's host
    restatement + the C oracle: both lengths identical, both costs within 1e-5), and through the property of
    tests/test_gpu_parity.py::test_mixed_locus_batch_at_per_gpu_size: reads that are copies of one clean template agree on the
    allele up to noise."""
    import filecmp
    import shutil
    import tempfile

    import pandas as pd
    import torch

    from oracle import oracle
    from warpstr_amd.signal_prep import process_raw
    from warpstr_amd.wrapper import main_wrapper_loci
    root = tempfile.mkdtemp(prefix='wsx_cfg5_full_', dir=scratch_dir())
    try:
        device = torch.device('cuda', local)
        specs = [(f'locus{i}', p, cfg5_flank(p, 11 + i), (500, 5000), 11 + i) for i, p in enumerate(CFG5_PATTERNS)]
        t0 = time.perf_counter()
        info = {}
        first, raws = make_locus_dirs(os.path.join(root, 'a'), specs, reads_per_locus, 79, device=device, info=info)
        second, _ = make_locus_dirs(os.path.join(root, 'b'), specs, reads_per_locus, 79, device=device)
        gen_s = time.perf_counter() - t0
        threads = min(8, os.cpu_count() or 1)
        tm0, tm = {}, {}
        main_wrapper_loci(first, threads, raw_reads=raws, device=local, quiet=True, timings=tm0)
        main_wrapper_loci(second, threads, raw_reads=raws, device=local, quiet=True, timings=tm)
        same = all(filecmp.cmp(os.path.join(a.path, rel), os.path.join(b.path, rel), shallow=False)
                   for a, b in zip(first, second) for rel in ('overview.csv', 'predictions/sequences/all.fasta'))
        n = len(second) * reads_per_locus
        rng = np.random.default_rng(5)
        bad, checked, called, samples, agree = [], 0, 0, 0, []
        for (name, pattern, fl, _, _), loc in zip(specs, second):
            df = pd.read_csv(os.path.join(loc.path, 'overview.csv'), dtype={'read_name': str})
            called += int((df['results'] >= 0).sum())
            pick, locus = info[name]
            oa = {False: oracle.Automaton.from_table(locus.template, fl), True: oracle.Automaton.from_table(locus.reverse, fl)}
            for i in rng.integers(0, len(df), size=12):
                raw = raws[df['read_name'][i]]
                o = oracle.call_read(oa[bool(df['reverse'][i])], process_raw(raw, (0, len(raw) - 1), 'Brute'), debug=False)
                ok = o.status == 0 and (int(df['orig'][i]), int(df['results'][i])) == (o.len1, o.len2)
                for a, b in ((float(df['dtw_cost1'][i]), o.cost1), (float(df['dtw_cost2'][i]), o.cost2)):
                    ok = ok and abs(a - b) <= 1e-5 * max(abs(b), 1e-300)
                checked += 1
                if not ok:
                    bad.append(f'{name}:{i}')
            l2 = df['results'].to_numpy()
            for k in np.unique(pick):
                v = l2[pick == k]
                agree.append(float(np.mean(np.abs(v - np.median(v)) <= 6)))
            samples += int(sum(len(raws[nm]) for nm in df['read_name']))
        return {'workload': f'BASELINE configs[4] at full size on one GPU: 8 loci x 2 strands x {reads_per_locus} reads = {n} reads, T in [500, 5000], '
                            f'S={min(min(l.template.n_states, l.reverse.n_states) for _, l in info.values())}..'
                            f'{max(max(l.template.n_states, l.reverse.n_states) for _, l in info.values())} states, one handle, '
                            'main_wrapper_loci from raw int16 reads in host memory to the output files',
                'reads': n, 'samples': samples, 'called_ok': called, 'reads_per_s': n / tm['total_s'], 'wall_s': tm['total_s'],
                'first_run_wall_s': tm0['total_s'], 'generation_s': gen_s, 'driver': _driver_timings(tm, len(second)),
                'workspace_bytes': tm.get('workspace_bytes'), 'workspace_bytes_per_sample_of_the_run': (tm.get('workspace_bytes') or 0) / max(samples, 1),
                'deterministic': {'runs': 2, 'files_identical': bool(same)},
                'verified': {'reads': checked, 'mismatches': len(bad), 'first_mismatches': bad[:5],
                             'fields': 'orig, results identical; dtw_cost1, dtw_cost2 within 1e-5 relative', 'against': 'oracle/ (CPU)'},
                'copies_of_one_template_within_6_bases_of_their_median': float(np.mean(agree))}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def vbz_kernel_leg(path, local, n_blocks=2048, launches=20):
    """wsx_vbz_decode alone at the size of a from_fast5 batch: n_blocks blocks cycling through the upstream test file's ten reads
    (their real StreamVByte blocks, zstd undone), HIP-event time per launch on the handle's stream against the bytes a launch has
    to move (block bytes in + 2 B per sample out); the first ten blocks checked against oracle/vbz.py."""
    import ctypes as C
    import struct

    import torch

    from oracle import vbz as ovbz
    from warpstr_amd import _lib, fast5, synth
    from warpstr_amd.caller import HipCaller
    h, zs = fast5._libs()
    real = []
    with fast5.Fast5File(path) as f:
        for rid in f.read_ids():
            d, n, prm, chunk_len = f._open_signal(rid)
            try:
                for _, _, buf, size, plain in f._chunks(d, n, chunk_len):
                    assert not plain and prm[:2] == [0, 2] and struct.unpack_from('<I', buf, 0)[0] == 2 * n
                    body = bytes(buf[4:size])
                    m = zs.ZSTD_getFrameContentSize(body, len(body))
                    blk = C.create_string_buffer(m)
                    assert zs.ZSTD_decompress(blk, m, body, len(body)) == m
                    real.append((np.frombuffer(blk.raw[:m], np.uint8), n, bool(prm[2])))
            finally:
                h.H5Dclose(d)
    locus = synth.make_locus('(AGC)', 16, 1)
    dev = torch.device('cuda', local)
    stream = torch.cuda.Stream(device=dev)
    hip = HipCaller([locus.template, locus.reverse], [16, 16], device=local, stream=stream.cuda_stream)
    blobs = [real[i % len(real)] for i in range(n_blocks)]
    src = np.concatenate([np.concatenate([b[0], np.zeros(-len(b[0]) % 16, np.uint8)]) for b in blobs])
    blocks = np.zeros(n_blocks, _lib.VBZ_BLOCK_DTYPE)
    at = out = 0
    for i, (blk, n, zz) in enumerate(blobs):
        blocks[i] = (at, len(blk), out, n, _lib.VBZ_SVB_ZIGZAG if zz else _lib.VBZ_SVB, n, 0)
        at += len(blk) + (-len(blk) % 16)
        out += n
    with torch.cuda.stream(stream):
        src_d = torch.from_numpy(src).to(dev)
        dst_d = torch.empty(out, dtype=torch.int16, device=dev)
        st_d = torch.empty(n_blocks, dtype=torch.int32, device=dev)
        args = (src_d.data_ptr(), len(src), blocks, dst_d.data_ptr(), out, st_d.data_ptr())
        for _ in range(3):
            hip.vbz_decode_device(*args)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
        ev[0].record()
        for k in range(launches):
            hip.vbz_decode_device(*args)
            ev[k + 1].record()
        stream.synchronize()
    ms = float(np.mean([ev[k].elapsed_time(ev[k + 1]) for k in range(launches)]))
    got = dst_d.cpu().numpy()
    ok = int(st_d.sum()) == 0 and all(np.array_equal(got[blocks[i]['dst_offset']:blocks[i]['dst_offset'] + blocks[i]['n_samples']],
                                                     ovbz.decode_block(*blobs[i])) for i in range(min(10, n_blocks)))
    hip.close()
    algo = int(blocks['src_bytes'].sum()) + 2 * out
    return {'kernel': 'vbz_decode_kernel', 'blocks_per_launch': n_blocks, 'samples_per_launch': out, 'algorithmic_bytes_per_launch': algo,
            'launch_ms': ms, 'roofline': {'bound': 'hbm', 'achieved': algo / ms / 1e6, 'peak': 8000.0, 'unit': 'GB/s', 'frac': algo / ms / 1e6 / 8000.0,
                                             'traffic': None},   # (counters: profiles/r05_vbz_pmc.log -- 1.03 x the algorithmic bytes)
            'samples_per_s': out / ms * 1e3, 'equal_to_oracle': bool(ok),
            'note': 'a workgroup per block, two workgroup scans per 2 048 values; bound by the instructions it issues (VALU half busy); the from_fast5 leg needs 9 x 10^9 samples/s of it'}


def zstd_kernel_leg(path, local, n_frames=2048, launches=10):
    """wsx_zstd_decode alone at the size of a from_fast5 batch: n_frames frames cycling through the upstream test file's ten chunks
    (their real zstd frames), HIP-event time per launch on the handle's stream; its output checked against libzstd for the first
    ten frames.  The bytes it has to move: the frames in, their content out (and once more through the literal area)."""
    import ctypes as C

    import torch

    from warpstr_amd import _lib, fast5, synth
    from warpstr_amd.caller import HipCaller
    h, zs = fast5._libs()
    real = []
    with fast5.Fast5File(path) as f:
        for rid in f.read_ids():
            d, n, prm, chunk_len = f._open_signal(rid)
            try:
                for _, _, buf, size, plain in f._chunks(d, n, chunk_len):
                    frame = bytes(buf[4:size])
                    real.append((np.frombuffer(frame, np.uint8), int(zs.ZSTD_getFrameContentSize(frame, len(frame)))))
            finally:
                h.H5Dclose(d)
    locus = synth.make_locus('(AGC)', 16, 1)
    dev = torch.device('cuda', local)
    stream = torch.cuda.Stream(device=dev)
    hip = HipCaller([locus.template, locus.reverse], [16, 16], device=local, stream=stream.cuda_stream)
    blobs = [real[i % len(real)] for i in range(n_frames)]
    table = np.zeros(n_frames, _lib.ZSTD_FRAME_DTYPE)
    at = out = 0
    parts = []
    for i, (fr, m) in enumerate(blobs):
        pad = -len(fr) % 16
        parts += [fr, np.zeros(pad, np.uint8)]
        table[i] = (at, len(fr), out, m)
        at += len(fr) + pad
        out += m + (-m % 16)
    src = np.concatenate(parts)
    with torch.cuda.stream(stream):
        src_d = torch.from_numpy(src).to(dev)
        dst_d = torch.empty(out, dtype=torch.uint8, device=dev)
        scr_d = torch.empty(out, dtype=torch.uint8, device=dev)
        st_d = torch.empty(n_frames, dtype=torch.int32, device=dev)
        args = (src_d.data_ptr(), len(src), table, dst_d.data_ptr(), out, scr_d.data_ptr(), st_d.data_ptr())
        for _ in range(2):
            hip.zstd_decode_device(*args)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
        ev[0].record()
        for k in range(launches):
            hip.zstd_decode_device(*args)
            ev[k + 1].record()
        stream.synchronize()
    ms = float(np.mean([ev[k].elapsed_time(ev[k + 1]) for k in range(launches)]))
    ok = int(st_d.sum()) == 0
    got = dst_d.cpu().numpy()
    for i in range(min(10, n_frames)):
        fr, m = blobs[i]
        ref = np.empty(m, np.uint8)
        ok = ok and zs.ZSTD_decompress(ref.ctypes.data, m, fr.tobytes(), len(fr)) == m and np.array_equal(got[int(table[i]['dst_offset']):][:m], ref)
    hip.close()
    frames_b, content_b = int(table['src_bytes'].sum()), int(table['dst_bytes'].sum())
    algo = frames_b + content_b
    return {'kernels': ['zstd_index_kernel', 'zstd_order_kernel', 'zstd_literals_kernel', 'zstd_sequences_kernel'], 'frames_per_launch': n_frames, 'frame_bytes_per_launch': frames_b,
            'content_bytes_per_launch': content_b, 'algorithmic_bytes_per_launch': algo, 'launch_ms': ms,
            'roofline': {'bound': 'hbm', 'achieved': algo / ms / 1e6, 'peak': 8000.0, 'unit': 'GB/s', 'frac': algo / ms / 1e6 / 8000.0, 'traffic': None},
            'frames_per_s': n_frames / ms * 1e3, 'equal_to_libzstd': bool(ok),
            'note': 'a Huffman stream is a chain of dependent table look-ups (two symbols a look-up where both codes fit 11 bits): four blocks to a '
                    'wavefront, the largest blocks first -- a launch is as long as one full block\'s chain; bound by the latency of that chain and by the LDS the '
                    'tables take, not by HBM; libzstd takes 0.06 ms of one core per frame '
                    '(2 048 frames: 123 ms of CPU, 7.7 ms on the 16 cores of the box)'}


def from_fast5_leg(n_copies, local):
    """The path real input takes: .fast5 files on disk -> output files (upstream: get_workload opens one fast5 per read through
    Fast5.get_data_processed, src/caller/wrapper.py:44-54, src/schemas/fast5.py:45-57).  n_copies copies of the upstream test
    file (tests/golden/real/batch_0.fast5: 10 real VBZ-compressed R9.4 reads of 59-170 k samples) laid out as the caller-only
    input (prepare_caller_only.py: an overview per locus whose `fast5_path` column points at the read's multi-read file), one
    locus -- the upstream test locus, (AAAT) at flank 110 -- per copy, through main_wrapper_loci to every locus's output files:
    once with the files read in this process, once on 16 reader processes (libhdf5 is not thread-safe); either way the readers
    undo zstd into page-locked arenas the upload starts from, and StreamVByte, zig-zag and the running sum are undone on the
    device (wsx_vbz_decode).  The genotype of the outputs must be the README's (44, 40)."""
    import contextlib
    import io
    import json as js
    import shutil
    import tempfile

    import pandas as pd

    from warpstr_amd import fast5, overview as ov
    from warpstr_amd.genotyper import run_genotyping_overview
    from warpstr_amd.wrapper import LocusPath, main_wrapper_loci
    real = os.path.join(ROOT, 'tests', 'golden', 'real')
    fj = js.load(open(os.path.join(real, 'flanks.json')))
    ex = pd.read_csv(os.path.join(real, 'example.csv'), dtype={'read_name': str})
    size = os.path.getsize(os.path.join(real, 'batch_0.fast5'))
    base = scratch_dir()
    if base is not None:   # room for the copies AND the staging buffers?
        st = os.statvfs(base)
        if st.f_bavail * st.f_frsize < n_copies * size * 3 + (4 << 30):
            base = None
    root = tempfile.mkdtemp(prefix='wsx_from_fast5_', dir=base)
    try:
        csv_head = 'read_name,fast5_path,reverse,l_start_raw,r_end_raw,run_id,saved\n'
        csv_rows = [f'{nm},%s,{bool(rv)},{int(a)},{int(b)},run_0,1\n' for nm, rv, a, b in
                    zip(ex['read_name'], ex['reverse'].astype(bool), ex['l_start_raw'], ex['r_end_raw'])]
        flanks = [fj['left_template'], fj['right_template'], fj['left_reverse'], fj['right_reverse']]

        def make(tag, n=n_copies, per_locus=1):
            """n copies of the file; a locus per `per_locus` copies (its overview lists the reads of all of them).  The tables are
            written as text (the rows pandas would write for these columns: 6 000 loci in a second instead of ten)."""
            loci = []
            for i0 in range(0, n, per_locus):
                loc = os.path.join(root, tag, f'copy{i0:04d}')
                text = [csv_head]
                for i in range(i0, min(n, i0 + per_locus)):
                    f5 = os.path.join(root, 'fast5', f'batch_{i:04d}.fast5')
                    if not os.path.exists(f5):
                        os.makedirs(os.path.dirname(f5), exist_ok=True)
                        shutil.copyfile(os.path.join(real, 'batch_0.fast5'), f5)
                    text += [row % f5 for row in csv_rows]
                ov.store_flanks(loc, flanks)
                with open(os.path.join(loc, 'overview.csv'), 'w') as f:
                    f.writelines(text)
                loci.append(LocusPath(loc, fj['sequence'], int(fj['flank_length']), f'copy{i0:04d}'))
            return loci
        warm = make('warm', 8)
        main_wrapper_loci(warm, 1, device=local, quiet=True)
        out = {'workload': f'{n_copies} copies of the upstream test fast5 (10 real VBZ reads each, {size} bytes), one (AAAT) flank-110 locus per copy, '
                           f'caller-only layout, files under {base or "the default temporary directory"} (page cache) -> output files',
               'reads': n_copies * len(ex)}
        many = min(16, os.cpu_count() or 1)
        from warpstr_amd.loci import cpu_share, default_readers
        out['cpus'] = {'visible': os.cpu_count(), 'usable_under_the_cgroup_quota': cpu_share()}
        knee = default_readers(many, True)   # reader processes of a run with sixteen host threads (the knee of reader_sweep below)
        legs = [('one_process', 1, None, n_copies, 1), ('reader_processes', many, None, n_copies, 1),
                # configs[4]'s shape from files: the same reads as eight loci (1 875 reads each at the default size)
                ('reader_processes_eight_loci', many, None, n_copies, -(-n_copies // 8))]
        st = os.statvfs(root)
        if st.f_bavail * st.f_frsize > 4 * n_copies * size * 3 + (4 << 30):   # (the run's fixed parts -- set-up, handle, the last
            legs.append(('reader_processes_4x_the_copies', many, None, 4 * n_copies, 1))   # batch's tail -- weigh less on a longer run)
            # the reader count swept on the long run (60 000 reads at the default size): same host threads, 16 ... 128 readers
            # ... and, on request, a run four times as long again (240 000 reads at the default size, 25 GB of copies, 16 s of the bench):
            # 91 k reads/s -- 24 000 loci take 0.6 s to set up, the readers 0.08 ms a read with four times the files (profiles/r06_from_fast5_240k.json)
            if st.f_bavail * st.f_frsize > 16 * n_copies * size * 2 + (16 << 30) and os.environ.get('WARPSTR_BENCH_LONG_FAST5'):
                legs.append(('reader_processes_16x_the_copies', many, None, 16 * n_copies, 1))
            sweep = [int(r) for r in os.environ.get('WARPSTR_BENCH_READER_SWEEP', '4,8,16,32,64,128').split(',')]
            legs += [(f'reader_sweep.{r}', many, r, 4 * n_copies, 1) for r in sweep if r <= (os.cpu_count() or 1)]
        only = os.environ.get('WARPSTR_BENCH_FAST5_ONLY')   # (a profiler run wants one leg: e.g. one_process -- no child processes)
        for tag, threads, readers, n, per_locus in [leg for leg in legs if not only or leg[0] == only or leg[0].startswith(only + '.')]:
            loci = make(tag, n, per_locus)
            n_reads = n * len(ex)
            tm = {'timeline': []} if os.environ.get('WARPSTR_BENCH_TIMELINE') else {}   # (the run's events, for scripts/exp_from_fast5.py)
            t_call = time.perf_counter()
            tables = main_wrapper_loci(loci, threads, readers=readers, device=local, quiet=True, timings=tm)
            call_s = time.perf_counter() - t_call   # (with what follows the outputs: the reader processes sent home, the handle closed)
            with contextlib.redirect_stdout(io.StringIO()):
                calls = [run_genotyping_overview(None, l.path, None).alleles for l in (loci[0], loci[-1])]
            lens = [tuple(int(v) for v in pd.read_csv(os.path.join(l.path, 'overview.csv'))['results'][:len(ex)]) for l in loci[::max(1, n // n_copies)]]
            rec = {'reads': n_reads, 'loci': len(loci), 'reads_per_s': n_reads / tm['total_s'], 'wall_s': tm['total_s'], 'call_returns_after_s': call_s,
                   'host_threads': tm.get('host_threads'),
                   'reader_processes': tm.get('reader_processes'), 'raw_MB': tm.get('raw_bytes', 0) / 1e6, 'uploaded_MB': tm.get('uploaded_bytes', 0) / 1e6,
                   'reader_mode': tm.get('reader_mode'), 'inside_wsx_caller_create': tm.get('handle_create_s'),
                   'inside_submit_upload': tm.get('submit_parts_s'), 'inside_wait_for_gpu': tm.get('collect_parts_s'),
                   'phases_s': {'setup': tm.get('setup_wall_s'), 'handle': tm['handle_s'], 'read_total': tm['read_s'],
                                'read_probe_lengths': tm.get('probe_s'), 'read_decode_into_staging': tm.get('decode_s'),
                                'decode_summed_over_reader_processes': tm.get('decode_worker_s'),
                                'submit_upload': tm['submit_s'], 'wait_for_gpu': tm['collect_s'], 'outputs': tm['store_s']},
                   'shared_staging_refused': tm.get('shared_staging_refused'),
                   'genotype_first_last': [list(c) for c in calls], 'all_loci_equal': bool(len(set(lens)) == 1)}
            if tm.get('timeline'):
                rec['timeline'] = [f'{t:8.4f} {name}' for name, t in tm['timeline']]
            if tag == 'reader_processes_16x_the_copies':   # (25 GB of copies: gone before the sweep makes its own)
                shutil.rmtree(os.path.join(root, tag), ignore_errors=True)
                for i in range(4 * n_copies, n):
                    try:
                        os.unlink(os.path.join(root, 'fast5', f'batch_{i:04d}.fast5'))
                    except OSError:
                        pass
            if tag.startswith('reader_sweep.'):
                out.setdefault('reader_sweep', {'reads': n_reads, 'default_readers_at_16_threads': knee})[tag.split('.')[1]] = {
                    k: rec[k] for k in ('reads_per_s', 'wall_s', 'call_returns_after_s', 'reader_processes', 'phases_s', 'genotype_first_last', 'all_loci_equal', 'timeline') if k in rec}
                shutil.rmtree(os.path.join(root, tag), ignore_errors=True)
            else:
                out[tag] = rec
        # where a read's time goes in one process, COLD -- every read of a run is read exactly once, from a file whose metadata
        # libhdf5 has not parsed yet: 40 fresh copies x 10 reads, each library call timed (ms per read)
        import ctypes as C
        h, zs = fast5._libs()
        native = fast5._vbz_native()
        cold = os.path.join(root, 'cold')
        os.makedirs(cold)
        t = {k: 0.0 for k in ('H5Fopen', 'H5Dopen2_by_name', 'dataspace_and_filter_queries', 'H5Dget_chunk_storage_size', 'H5Dread_chunk',
                              'zstd_decompress', 'streamvbyte_zigzag_prefix_sum', 'H5Dclose', 'H5Fclose')}
        ids = fast5.Fast5File(os.path.join(real, 'batch_0.fast5')).read_ids()
        n_cold, samples, chunk_bytes = 0, 0, 0
        clock = time.perf_counter
        for i in range(40):
            path = os.path.join(cold, f'c{i}.fast5')
            shutil.copyfile(os.path.join(real, 'batch_0.fast5'), path)
            t0 = clock()
            fid = h.H5Fopen(path.encode(), 0, 0)
            t['H5Fopen'] += clock() - t0
            for rid in ids:
                t0 = clock()
                d = h.H5Dopen2(fid, f'read_{rid}/Raw/Signal'.encode(), 0)
                t1 = clock()
                sp = h.H5Dget_space(d)
                ns = h.H5Sget_simple_extent_npoints(sp)
                h.H5Sclose(sp)
                pl = h.H5Dget_create_plist(d)
                cd, ne, flags, fc, name = (C.c_uint * 8)(), C.c_size_t(8), C.c_uint(), C.c_uint(), C.create_string_buffer(64)
                h.H5Pget_filter2(pl, 0, C.byref(flags), C.byref(ne), cd, 64, name, C.byref(fc))
                cl = (C.c_uint64 * 1)(0)
                h.H5Pget_chunk(pl, 1, cl)
                h.H5Pclose(pl)
                t2 = clock()
                off, sz, mask = (C.c_uint64 * 1)(0), C.c_uint64(), C.c_uint32()
                h.H5Dget_chunk_storage_size(d, off, C.byref(sz))
                t3 = clock()
                buf = C.create_string_buffer(sz.value)
                h.H5Dread_chunk(d, 0, off, C.byref(mask), buf)
                t4 = clock()
                zsz = zs.ZSTD_getFrameContentSize(C.cast(C.addressof(buf) + 4, C.c_char_p), sz.value - 4)
                zout = C.create_string_buffer(zsz)
                zs.ZSTD_decompress(zout, zsz, C.cast(C.addressof(buf) + 4, C.c_char_p), sz.value - 4)
                t5 = clock()
                dst = np.empty(ns, np.int16)
                if native is not None:   # the whole chunk decoder, minus the zstd share measured above
                    native[0](buf, sz.value, int(cd[2]), int(cd[3]), native[1], native[2], dst.ctypes.data, ns)
                t6 = clock()
                h.H5Dclose(d)
                t7 = clock()
                for key, dt in zip(('H5Dopen2_by_name', 'dataspace_and_filter_queries', 'H5Dget_chunk_storage_size', 'H5Dread_chunk', 'zstd_decompress',
                                    'H5Dclose'), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t7 - t6)):
                    t[key] += dt
                t['streamvbyte_zigzag_prefix_sum'] += max((t6 - t5) - (t5 - t4), 0.0)
                n_cold += 1
                samples += int(ns)
                chunk_bytes += int(sz.value)
            t0 = clock()
            h.H5Fclose(fid)
            t['H5Fclose'] += clock() - t0
        t = {k: v / n_cold * 1e3 for k, v in t.items()}
        t['total'] = sum(t.values())
        t['mean_samples'], t['mean_chunk_bytes'], t['reads'] = samples / n_cold, chunk_bytes / n_cold, n_cold
        t['note'] = ('cold files: each read is opened once; H5Fopen / H5Fclose are per FILE of ten reads, shown per read; the decoder '
                     'call includes a second zstd pass, subtracted')
        out['per_read_ms_one_process_cold'] = t
        out['vbz_decode_kernel'] = vbz_kernel_leg(os.path.join(real, 'batch_0.fast5'), local)
        out['zstd_decode_kernels'] = zstd_kernel_leg(os.path.join(real, 'batch_0.fast5'), local)
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


def launch_ranks(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher around it (WORLD_SIZE unset): start the N ranks as a CHILD
    `python -m torch.distributed.run` of this process -- which has not touched the GPU and never will -- and relay rank 0's JSON
    line and the job's exit code.  Returns the exit code."""
    import socket
    import subprocess
    backend = os.environ.get('WARPSTR_BENCH_BACKEND', 'nccl')
    if backend == 'nccl':
        import torch
        have = torch.cuda.device_count()   # (counts devices without creating a HIP context on this image)
        if have < n_gpus:
            print(f'bench.py: --gpus {n_gpus} over RCCL needs {n_gpus} GPUs on this node, {have} visible '
                  '(one rank per GPU; WARPSTR_BENCH_BACKEND=gloo runs the ranks on the GPUs there are, as a rehearsal)', file=sys.stderr)
            return 2
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, WARPSTR_BENCH_LAUNCHED_BY=str(os.getpid()))
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // n_gpus)))   # (torchrun would set 1 and say so)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    out, _ = proc.communicate()
    lines = out.decode('utf-8', 'replace').splitlines()
    line = next((l for l in reversed(lines) if l.startswith('{') and '"metric"' in l), None)
    for l in lines:
        if l is not line:
            print(l, file=sys.stderr)
    if line is not None:
        sys.stdout.write(line + '\n')
        sys.stdout.flush()
    elif proc.returncode == 0:
        print('bench.py: the ranks ended without a result line', file=sys.stderr)
        return 4
    return proc.returncode


def optional_leg(leg, *a):
    """A leg above the kernels (the product driver) must not take the headline line down with it: an
    exception becomes {'failed': ...} in its place (a result that DIFFERS still fails the run: the callers check that)."""
    try:
        return leg(*a)
    except Exception as e:  # noqa: BLE001
        import traceback
        print(f'bench.py: {leg.__name__} failed:\n{traceback.format_exc()}', file=sys.stderr)
        return {'failed': f'{type(e).__name__}: {e}'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', choices=['headline', 'cfg1', 'cfg5'], default='headline')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
    ap.add_argument('--reads', type=int, default=0, help='reads per GPU (weak) / in total (strong); default by workload')
    ap.add_argument('--samples', type=int, default=2000, help='samples per read (headline workload)')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the 15 s CPU baseline (a 256-read check remains)')
    ap.add_argument('--no-verify', action='store_true')
    ap.add_argument('--no-secondary', action='store_true',
                    help='the default single-GPU headline run also times cfg1, cfg5 and the path from raw int16 segments '
                         '(reported under "secondary"); this switch leaves them out')
    ap.add_argument('--workspace-limit-gib', type=float, default=0.0,
                    help='wsx_caller_set_workspace_limit for the main handle (default: the library chooses from the free device memory)')
    ap.add_argument('--from-fast5', type=int, default=1500, help='copies of the upstream test fast5 in the from_fast5 leg of the default run (0: leave it out)')
    ap.add_argument('--cfg5-full', type=int, default=50000, help='reads per locus of the cfg5_full leg of the default run: configs[4] at its full '
                                                               'size, 8 loci x that many reads, on the one GPU (0: leave it out)')
    ap.add_argument('--many-loci', type=int, default=2000, help='loci of the many_loci leg of the default run (0: leave it out)')
    ap.add_argument('--from-raw', action='store_true',
                    help='also time the path from raw int16 segments (host and HBM resident): loader kernels + caller with the '
                         'called sequences requested; reported as from_raw next to the headline')
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be at least 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # not under a launcher: this process becomes the launcher (before torch or HIP is loaded) and the ranks its children
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

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
        raise SystemExit(f'--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: start bench.py with --gpus {world}, '
                         'or without a launcher (it starts its own ranks)')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU path)')
    # one process per GPU; WARPSTR_BENCH_BACKEND=gloo lets the multi-rank logic be exercised on a 1-GPU box
    backend = os.environ.get('WARPSTR_BENCH_BACKEND', 'nccl')
    if backend == 'nccl' and world > torch.cuda.device_count():
        raise SystemExit(f'bench.py: {world} ranks over RCCL need {world} GPUs on this node, {torch.cuda.device_count()} visible (one rank per GPU)')
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
    from warpstr_amd.dist import gather_results, shard_reads

    # ---- the workload of this rank ------------------------------------------------------------------------------
    default_reads = {'headline': 100000, 'cfg1': 20000, 'cfg5': 50000}[args.workload]
    n_arg = args.reads or default_reads
    strong = args.scaling == 'strong'
    shards = None
    if strong:
        # configs[3]: ONE workload of n_arg reads; rank r calls the reads shard_reads() gives it (all ranks compute the
        # same partition from the same seeded global description), the all-gather returns every record to every rank
        n_total = n_arg
        if args.workload != 'headline':
            raise SystemExit('--scaling strong is defined for the headline workload (configs[3])')
        gpicks = np.random.default_rng(1000).integers(0, 2048, size=n_total)
        shards = shard_reads(np.full(n_total, args.samples, np.int64), world)
        wl = make_headline(len(shards[rank]), args.samples, 1000 + rank, device, picks=gpicks[shards[rank]])
    else:
        n_total = n_arg * world
        if args.workload == 'headline':
            wl = make_headline(n_arg, args.samples, 1000 + rank, device)
        elif args.workload == 'cfg1':
            pat, fl, tr = CFG1
            wl = make_ragged('cfg1', [(pat, fl, tr, 1, None)], n_arg, 1000 + rank, device)
        else:
            wl = make_ragged('cfg5', [(p, cfg5_flank(p, 11 + i), (500, 5000), 11 + i, None) for i, p in enumerate(CFG5_PATTERNS)],
                             n_arg, 1000 + rank, device)
    n = wl.n
    n_pad = (n_total + world - 1) // world if strong else n  # all_gather_into_tensor wants equal shards
    stream = torch.cuda.current_stream().cuda_stream
    hip = HipCaller(wl.tables, wl.flanks, device=local, stream=stream,  # the library's defaults unless asked otherwise
                    workspace_limit=int(args.workspace_limit_gib * (1 << 30)) or None)
    # Result buffers: the all-gather of step k runs on a side stream while the kernels of step k+1 already fill the
    # next buffer (the collective moves 56 B per read and rank over xGMI: latency-bound, nothing for the CUs to do).
    res_bufs = [torch.zeros((n_pad, _lib.RESULT_DTYPE.itemsize), dtype=torch.uint8, device=device) for _ in range(N_BUF)]
    gather_stream = torch.cuda.Stream(device=device) if collective else None
    gather_done = [None] * N_BUF
    step_no = [0]
    # back-to-back steps: a call no longer drains into the handle's stream, the next step's chunks follow on the
    # library's internal streams (wsx_caller_set_pipelined); consumers are ordered after a step with join()
    hip.set_pipelined(True)

    def step():
        k = step_no[0] % N_BUF
        step_no[0] += 1
        results = res_bufs[k]
        if gather_done[k] is not None:  # the gather that read this buffer N_BUF steps ago
            torch.cuda.current_stream().wait_event(gather_done[k])
        hip.call_device(wl.signal.data_ptr(), wl.offsets, wl.aut, results.data_ptr())
        if collective:
            hip.join(gather_stream.cuda_stream)  # every kernel of this step is ahead of the gather
            with torch.cuda.stream(gather_stream):
                out = torch.empty((world * n_pad, results.shape[1]), dtype=torch.uint8, device=device)
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
    hip.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    hip.timing_window(True)   # the fill kernels' HIP events of ALL timed steps are kept
    t0 = time.perf_counter()
    for _ in range(args.steps):
        allres = step()
    hip.synchronize()        # pipelined calls end on the library's own streams
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    last_k = (step_no[0] - 1) % N_BUF   # the buffer the last timed step's records lie in
    tm = hip.last_timing()  # HIP events on the launch streams, every fill launch of the timed region
    fb, fe, fr = hip.fill_intervals()
    hip.timing_window(False)
    workspace = hip.workspace()  # what the timed steps needed (the single-stream step below is one 100k-read chunk: bigger)
    workspace['limit_bytes'] = hip.workspace_limit()
    # untimed extra step on ONE stream: the fill kernel's duration when nothing runs beside it (VALU roofline)
    hip.set_streams(1)
    step()
    hip.synchronize()
    torch.cuda.synchronize()
    ab, ae, _ = hip.fill_intervals()
    hip.set_streams(4)
    # what proves the ranks and the collective were real: every rank's device, clock and a checksum of the records it computed in
    # the last timed step, all-gathered as objects; rank 0 compares the checksums with those of the slices the step's own
    # all-gather (RCCL: all_gather_into_tensor on the gather stream) delivered
    ranks = None
    if world > 1 or self_gather:
        import zlib
        props = torch.cuda.get_device_properties(local)
        own_n = len(shards[rank]) if strong else n
        me = {'rank': rank, 'local_rank': int(os.environ.get('LOCAL_RANK', '0')), 'device': local, 'device_name': props.name,
              'pci_bus_id': getattr(props, 'pci_bus_id', None), 'uuid': str(getattr(props, 'uuid', '')) or None, 'pid': os.getpid(),
              'reads': own_n, 'ms_per_step': dt / args.steps * 1e3,
              'records_crc32': zlib.crc32(res_bufs[last_k][:own_n].cpu().numpy().tobytes())}
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, me)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    rc = 0
    if rank == 0:
        table = allres.cpu().numpy().view(_lib.RESULT_DTYPE).reshape(world, n_pad)
        mine = table[0][:n]
        ok_all = int(sum((table[r][:(len(shards[r]) if strong else n)]['status'] == 0).sum() for r in range(world)))
        S = max(t.n_states for t in wl.tables)
        samples = float(wl.offsets[-1])
        reads_per_s = (n_total if strong else n * world) * args.steps / dt
        # dominant kernel = the DTW fill.  One launch = one pass over a chunk's reads of one kernel variant; durations are
        # HIP events recorded around each launch on its stream.  Launches of different chunks overlap (they run on
        # different streams), so the time the device spent filling is the UNION of the intervals, not their sum.
        launches_total = max(len(fb), 1)
        fill_union = union_ms(fb, fe)
        fill_sum = float((fe - fb).sum())
        # algorithmic bytes: 12T+32 B per read for both passes = 6T+16 per read and pass
        algo_bytes_total = samples * 6.0 * 2 * args.steps + 16.0 * float(fr.sum())
        # roofline.achieved = algorithmic bytes of a launch / the kernel's average launch duration (HIP events around every
        # fill launch of the timed region, on the launch's own stream): what `rocprofv3 --kernel-trace --stats` averages for
        # the same command (profiles/r03_*_kernel_stats.csv).  Launches of different chunks overlap, so their sum exceeds
        # the step; the figure over the UNION of the intervals rides along as achieved_over_union.
        achieved = algo_bytes_total / (fill_sum * 1e-3) / 1e9
        achieved_union = algo_bytes_total / (fill_union * 1e-3) / 1e9
        kernel = hip.kernel_name(0)
        kernels = sorted({hip.kernel_name(a) for a in range(len(wl.tables))})
        prof = fill_profile(kernel)
        stale = prof.get('stale') if prof else None
        if stale:
            prof = None
        traffic = None
        if prof is not None and prof.get('hbm_bytes_per_sample') is not None:  # HBM bytes per launch from the PMC passes, scaled to this launch size
            traffic = prof['hbm_bytes_per_sample'] * samples * 2 * args.steps / launches_total
        alone_ms = float((ae - ab).sum()) / max(len(ab), 1)
        alone_rows = samples * 2 / max(len(ab), 1)
        cells_per_s = samples * 2 * args.steps * S / (fill_union * 1e-3)
        if wl.name == 'headline':
            head = f'BASELINE configs[{3 if strong else 2}]: '
        elif wl.name == 'cfg5':
            head = 'BASELINE configs[4] at one GPU\'s share: '
        else:
            head = 'shape of the upstream test case (every flank-110 locus): '
        out = {
            'metric': 'reads/s (STR segments aligned)', 'value': reads_per_s, 'unit': 'reads/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': head + (f'{n_total} reads in total sharded over {world} GPU(s)' if strong else f'{n} reads/GPU') +
                                   f', {wl.desc}, both passes',
                       'name': wl.name, 'reads_per_gpu': n, 'reads_total': n_total, 'mean_samples_per_read': samples / n,
                       'states': S, 'called_ok': ok_all, 'workspace_limit_bytes': workspace['limit_bytes'],
                       'chunk_plan': {'chunks_per_call': launches_total / args.steps / 2.0 / max(len(kernels), 1), 'streams': 4,
                                      'pipelined_calls_in_flight': 2 if n >= 32768 else 4},
                       'results_gather': ((f'{backend} all_gather of 56-B records per step' +
                                          (', overlapped with the next step' if collective else ' (CPU test path, synchronous)')) if world > 1
                                          else ('one-rank nccl group (self test)' if self_gather else 'none (1 GPU)'))},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'launch_ms': fill_sum / launches_total, 'algorithmic_bytes_per_launch': algo_bytes_total / launches_total,
                         'achieved_over_union': achieved_union, 'frac_over_union': achieved_union / HBM_PEAK_GBPS,
                         'kernel': kernel, 'kernels': kernels,
                         'fill_union_ms_per_launch': fill_union / launches_total, 'launches_per_step': launches_total / args.steps,
                         'reads_per_launch': float(fr.sum()) / launches_total,
                         'fill_union_ms_per_step': fill_union / args.steps,
                         'note': 'achieved = algorithmic bytes per fill launch / launch_ms, the mean HIP-event duration of the '
                                 'fill launches of the timed region (= what rocprofv3 --stats averages for this command); '
                                 'launches of different chunks overlap on different streams, achieved_over_union divides by the '
                                 'union of their intervals instead. min-plus recurrence: bound by fp64 VALU issue and the LDS '
                                 'pipe, not HBM (see valu_roofline)'},
            'valu_roofline': (valu_roofline(prof, alone_ms, alone_rows, kernel) if prof is not None else
                              {'launch_ms_alone': alone_ms, 'counters': stale or 'none (profiling run)'}),
            'valu': {'dp_cells_per_s': cells_per_s},
            # first enqueue to last finish of the timed region on the device clock (HIP events), per step
            'device_ms_per_step': tm['total_ms'] / args.steps,
            # device memory the handle holds for this workload (all work sets of all streams), and per sample of a call
            'workspace': workspace,
        }
        if ranks is not None:
            import zlib
            seen = [zlib.crc32(table[r][:ranks[r]['reads']].tobytes()) for r in range(world)]
            out['ranks'] = {'world_size': dist.get_world_size(), 'backend': dist.get_backend(),
                            'launched_by': 'bench.py itself (child torch.distributed.run)' if os.environ.get('WARPSTR_BENCH_LAUNCHED_BY') else 'an outer launcher',
                            'per_rank': ranks, 'ms_per_step_max_over_ranks': dt / args.steps * 1e3,
                            'distinct_devices': len({(r['device'], r['pci_bus_id'], r['uuid']) for r in ranks}),
                            'gathered_records_equal_every_ranks_own': bool(all(seen[r] == ranks[r]['records_crc32'] for r in range(world)))}
            if not out['ranks']['gathered_records_equal_every_ranks_own']:
                rc = 3
                print('bench.py: the all-gathered records differ from what the ranks computed', file=sys.stderr)
        if not args.no_verify:
            nv = min(n, 4096 if not args.no_cpu_baseline else 256)
            sample = wl.signal[: int(wl.offsets[nv])].cpu().numpy()
            ores, base = oracle_sample(wl, sample, nv, 15.0 if not args.no_cpu_baseline else 2.0)
            if world == 1 and not args.no_cpu_baseline:
                out['cpu_baseline'] = base
            out['verified'] = verify(mine, ores)
            if out['verified']['mismatches']:
                rc = 3
                print(f"bench.py: {out['verified']['mismatches']} of {len(ores)} reads differ from the oracle", file=sys.stderr)
        if 'cpu_baseline' in out:
            # the reference's own Python caller cannot travel to the GPU box; its timing on this shape was taken in the
            # development container (Pool(8), 8 vCPUs) and rides along as a constant
            out['cpu_baseline']['reference_python'] = {'value': 12.1, 'unit': 'reads/s', 'cores': 8, 'kind': 'reference',
                                                       'sample': 'upstream WarpSTR.run through Pool(8) on 2 kSample x 64-state '
                                                                 'reads, development container (BASELINE.md section 2); a constant, '
                                                                 'not measured in this run'}
        secondary = (world == 1 and wl.name == 'headline' and args.reads == 0 and not args.no_secondary and not args.no_verify
                     and not os.environ.get('WARPSTR_BENCH_PROFILING'))
        if (args.from_raw or secondary) and world == 1:
            out['from_raw'] = from_raw_leg(hip, wl, device, max(3, args.steps // 2), 2)
            if not out['from_raw']['identical_to_f64_path']['identical']:
                rc = 3
                print('bench.py: the from-raw path and the float64 path disagree', file=sys.stderr)
        if secondary:
            # driver-timed numbers for the production shape and configs[4]'s share, each checked against the oracle
            hip.close()
            del hip
            out['secondary'] = {}
            pat, fl, tr = CFG1
            legs = (('cfg1', lambda: make_ragged('cfg1', [(pat, fl, tr, 1, None)], 20000, 1000, device)),
                    ('cfg5', lambda: make_ragged('cfg5', [(p, cfg5_flank(p, 11 + i), (500, 5000), 11 + i, None)
                                                          for i, p in enumerate(CFG5_PATTERNS)], 50000, 1000, device)))
            for name, make in legs:
                leg = secondary_leg(make(), local, device, args.steps, max(args.warmup, 4), 256)
                out['secondary'][name] = leg
                if leg['verified']['mismatches']:
                    rc = 3
                    print(f"bench.py: secondary workload {name}: {leg['verified']['mismatches']} reads differ from the oracle", file=sys.stderr)
            out['secondary']['from_raw'] = out['from_raw']
            # the product seam above the kernels: configs[4]'s share and the many-loci regime through main_wrapper_loci
            out['secondary']['cfg5']['through_driver'] = optional_leg(cfg5_driver_leg, 6250, local)
            if args.cfg5_full > 0:
                full = out['secondary']['cfg5_full'] = optional_leg(cfg5_full_leg, args.cfg5_full, local)
                if full.get('verified', {}).get('mismatches') or not full.get('deterministic', {'files_identical': True})['files_identical']:
                    rc = 3
                    print('bench.py: cfg5_full: results differ from the oracle or between two runs', file=sys.stderr)
            if args.from_fast5 > 0:
                out['from_fast5'] = optional_leg(from_fast5_leg, args.from_fast5, local)
            if args.many_loci > 0:
                out['many_loci'] = optional_leg(many_loci_leg, args.many_loci, 30, min(32, args.many_loci), local)
                if not out['many_loci'].get('outputs_identical', {'identical': True})['identical']:
                    rc = 3
                    print('bench.py: many_loci: the batched driver and the per-locus loop wrote different files', file=sys.stderr)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if world > 1 or self_gather:
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == '__main__':
    main()

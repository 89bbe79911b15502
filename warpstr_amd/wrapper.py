"""main_wrapper: step 3 of the pipeline for one locus, on the GPU (src/caller/wrapper.py:17-54).

Reads <locus_path>/overview.csv and the flanks written by the squiggler step, calls every `saved` read
with the HIP caller, and writes the same outputs as the reference: overview.csv columns
(results, orig, dtw_cost1, dtw_cost2), predictions/sequences/*.fasta and, for loci with more than one
repeat unit, predictions/complexSTR_analysis/complex_repeat_units.csv.  Plots are not produced.
"""
import os
import sys
from typing import Callable, List, Optional

import numpy as np

from . import overview as ov
from .caller import CallerConfig, CallerWrapper, ReadSignal, RescalerConfig
from .fast5 import read_raw_signal
from .signal_prep import process_raw
from .units import break_into_units, collapse_repeats

FAST5_SUBDIR, ANNOT_SUBDIR = 'fast5', 'annot'


def _fast5_loader(spike_removal: str) -> Callable[[str, int, int], np.ndarray]:
    """Host-side loader (the restatement of Fast5.get_data_processed in signal_prep; main_wrapper prepares on the GPU)."""
    def load(path: str, l_start_raw: int, r_end_raw: int) -> np.ndarray:
        return process_raw(read_raw_signal(path), (l_start_raw, r_end_raw), spike_removal)
    return load


def annot_fast5_path(path: str, run_id, read_name: str) -> str:
    return os.path.join(path, FAST5_SUBDIR, str(run_id), ANNOT_SUBDIR, read_name + '.fast5')


def get_workload(df_overview, path: str, signal_loader: Callable[[str, int, int], np.ndarray]) -> List[ReadSignal]:
    """All `saved` rows of the overview, in overview order (src/caller/wrapper.py:44-54)."""
    work: List[ReadSignal] = []
    for row in df_overview.itertuples():
        if row.saved:
            fast5path = annot_fast5_path(path, row.run_id, row.Index)
            sig = signal_loader(fast5path, int(row.l_start_raw), int(row.r_end_raw))
            work.append(ReadSignal(row.Index, bool(row.reverse), np.asarray(sig, dtype=np.float64)))
    return work


def get_raw_workload(df_overview, path: str, raw_reader: Callable[[str], np.ndarray] = read_raw_signal):
    """The same rows as get_workload, as raw DAC reads + (l_start_raw, r_end_raw): input of CallerWrapper.run_raw."""
    names, reverses, raws, positions = [], [], [], []
    for row in df_overview.itertuples():
        if row.saved:
            names.append(row.Index)
            reverses.append(bool(row.reverse))
            fast5path = annot_fast5_path(path, row.run_id, row.Index)
            if raw_reader is read_raw_signal and not os.path.exists(fast5path) and hasattr(row, 'fast5_path'):
                raw = read_raw_signal(str(row.fast5_path), row.Index)  # caller-only input: read kept in its multi-read file
            else:
                raw = raw_reader(fast5path)
            raws.append(np.ascontiguousarray(raw, dtype=np.int16))
            positions.append((int(row.l_start_raw), int(row.r_end_raw)))
    return names, reverses, raws, positions


class LocusPath:
    """The three attributes of upstream's Locus (src/schemas/locus.py) that step 3 reads."""

    def __init__(self, path: str, sequence: str, flank_length: int, name: Optional[str] = None):
        self.path, self.sequence, self.flank_length = path, sequence.upper(), int(flank_length)
        self.name = name or os.path.basename(os.path.normpath(path))


def main_wrapper(locus, threads=1, flank_length: Optional[int] = None, *args,
                 caller_config: Optional[CallerConfig] = None, rescaler_config: Optional[RescalerConfig] = None,
                 signal_loader: Optional[Callable[[str, int, int], np.ndarray]] = None,
                 raw_reader: Callable[[str], np.ndarray] = read_raw_signal, device: int = 0, shard: bool = False, pore_model=None,
                 **kwargs):
    """Load data for calling and handle results (src/caller/wrapper.py:17-41).

    main_wrapper(locus, threads) -- the reference's call: `locus` is any object with `.path`, `.sequence`, `.flank_length`.
    main_wrapper(locus_path, sequence, flank_length[, threads]) -- the same without a Locus object.
    Returns (df_overview, df_collapsed) and writes what upstream writes except the plots: overview.csv columns,
    predictions/sequences/*.fasta, summaries/state_similarity.csv, and for loci with several repeat units
    predictions/complexSTR_analysis/complex_repeat_units.csv.

    shard=True: this process is one rank of a `python -m torch.distributed.run` job whose ranks ALL call main_wrapper for
    the same locus; the reads are dealt over the ranks (main_wrapper_loci).  It is an explicit argument -- a job whose ranks
    work on different loci calls with shard=False and no collective is entered."""
    if isinstance(locus, str):  # (locus_path, sequence, flank_length[, threads])
        locus = LocusPath(locus, threads, flank_length)
        threads = args[0] if args else kwargs.pop('threads', 1)
    if kwargs:
        raise TypeError(f'main_wrapper: unexpected arguments {sorted(kwargs)}')
    caller_config = caller_config or CallerConfig()
    if shard:
        return main_wrapper_loci([locus], threads, caller_config=caller_config, rescaler_config=rescaler_config,
                                 signal_loader=signal_loader, raw_reader=raw_reader, device=device, shard=True, pore_model=pore_model)[0]
    overview_path, df_overview = ov.load_overview(locus.path)
    cw = CallerWrapper(locus, threads, caller_config=caller_config, rescaler_config=rescaler_config, device=device, pore_model=pore_model)
    if signal_loader is None:
        # default: int16 reads straight from the .fast5 files, prepared on the GPU
        names, reverses, raws, positions = get_raw_workload(df_overview, locus.path, raw_reader)
        results = cw.run_raw(names, reverses, raws, positions, caller_config.spike_removal)
    else:
        workload = get_workload(df_overview, locus.path, signal_loader or _fast5_loader(caller_config.spike_removal))
        reverses = [w.reverse for w in workload]
        results = cw.run(workload)
    return _store_outputs(locus, overview_path, df_overview, results, reverses, write=True)


def main_wrapper_loci(loci, threads=1, **kwargs):
    """main_wrapper for every locus of a run through ONE handle (mixed-locus batches): warpstr_amd/loci.py."""
    from .loci import main_wrapper_loci as impl
    return impl(loci, threads, **kwargs)


def _store_outputs(locus, overview_path, df_overview, results, reverses, write: bool):
    """The results of all `saved` reads (overview order) -> overview columns, FASTA files, complex-unit table
    (src/caller/wrapper.py:24-41).  write=False: the tables are built and returned, the locus directory is not touched."""
    if hasattr(results, 'sequences'):  # a batch's CallerResults: the columns at once (a locus may have thousands of reads)
        seqs, resc = results.check().sequences()
        c1, c2 = results.costs()
        called = np.asarray(results.records['status']) == 0  # (a read that was not called: '' and NaN, as results[i] gives)
        c1, c2 = np.where(called, c1, np.nan), np.where(called, c2, np.nan)
        seq_results, cost_results = list(zip(seqs, resc)), list(zip(c1.tolist(), c2.tolist()))
    else:
        seq_results = [(r.seq, r.resc_seq) for r in results]
        cost_results = [(r.cost, r.resc_cost) for r in results]
    df_overview = ov.store_results(overview_path, df_overview, seq_results, cost_results, locus.path, write=write)
    df_collapsed = None
    units, repeat_units, offsets = break_into_units(locus.sequence.upper())
    if len(units) > 1:
        if write:
            print(f'Running complex genotyping as complex repeat units present: {units}')
        memo = {}  # the reads of a locus call a handful of distinct sequences: each is scanned once
        collapsed = []
        for s in seq_results:
            if s[1] not in memo:
                memo[s[1]] = collapse_repeats(s[1], repeat_units, offsets)
            collapsed.append(memo[s[1]])
        df_collapsed = ov.store_collapsed(collapsed, units, repeat_units, reverses, locus.path, write=write)
    return df_overview, df_collapsed


def prepare_caller_only(csv_path: str, output: str, base_dir: str = '.'):
    """Caller-only input (prepare_caller_only.py:41-112): a CSV with columns fast5_path, locus, read_name, reverse,
    l_start_raw, r_end_raw [, run_id] becomes <output>/<locus>/overview.csv (+ run_id 'run_0', saved 1).  The reference
    also copies every read into a single-read .fast5 under fast5/<run_id>/annot/; here the reads stay where they are
    (get_raw_workload follows the overview's fast5_path column), so nothing is rewritten.  Returns {locus: path}."""
    import pandas as pd
    df = pd.read_csv(csv_path, dtype={'read_name': str, 'locus': str, 'fast5_path': str})
    required = ['fast5_path', 'locus', 'read_name', 'reverse', 'l_start_raw', 'r_end_raw']
    if any(c not in df.columns for c in required) or any(c not in required + ['run_id'] for c in df.columns):
        raise ValueError(f'Not all required columns present in input CSV file. Required fields are: {required}')
    if 'run_id' not in df.columns:
        df['run_id'] = 'run_0'
    df['saved'] = 1
    df['fast5_path'] = [p if os.path.isabs(p) else os.path.abspath(os.path.join(base_dir, p)) for p in df['fast5_path']]
    out = {}
    for locus, part in df.groupby('locus', sort=False):
        for p, name in zip(part['fast5_path'], part['read_name']):
            if not os.path.exists(p):
                raise FileNotFoundError(f'fast5 file {p} of read {name} does not exist')
        locus_path = os.path.join(output, locus)
        os.makedirs(locus_path, exist_ok=True)
        part.to_csv(os.path.join(locus_path, ov.OVERVIEW_NAME), index=False)
        out[locus] = locus_path
    return out


def _npz_loader(path: str) -> Callable[[str, int, int], np.ndarray]:
    """Already normalised STR segments from an .npz archive (read name -> float64 array) instead of fast5 files."""
    archive = np.load(path)
    return lambda fast5path, l_start_raw, r_end_raw: archive[os.path.basename(fast5path)[:-len('.fast5')]]


def loci_from_config(cfg) -> list:
    """The loci of a parsed configuration (config.load_config) as LocusPath objects under <output>/<name>, the directory
    layout of upstream's Locus (src/schemas/locus.py:18-46)."""
    return [LocusPath(os.path.join(cfg.output, lc.name), lc.sequence, lc.flank_length, lc.name) for lc in cfg.loci]


def under_torchrun() -> bool:
    from . import dist as wdist
    return int(os.environ.get('WORLD_SIZE', '1')) > 1 or wdist.force_collectives()


def prepare_subdirs(locus_path: str, tr_region_calling: bool, force_overwrite: bool):
    """The directories of a locus that steps 3 and 4 write into, as upstream prepares them before a run
    (src/helpers.py:32-69: `summaries`; with tr_region_calling `predictions` and its DTW_alignments / basecalls / sequences, each
    emptied first when `force_overwrite` is set -- the previous results of the step go, nothing of the other steps is touched)."""
    import shutil

    def handle(path):
        if not os.path.exists(path):
            os.mkdir(path)
        elif force_overwrite:
            shutil.rmtree(path)
            os.mkdir(path)
    if not os.path.exists(locus_path):
        os.mkdir(locus_path)
    if not os.path.exists(os.path.join(locus_path, 'summaries')):
        os.mkdir(os.path.join(locus_path, 'summaries'))
    if tr_region_calling:
        pred = os.path.join(locus_path, ov.PREDICTIONS_SUBDIR)
        handle(pred)
        for sub in ('DTW_alignments', 'basecalls', 'sequences'):
            handle(os.path.join(pred, sub))


def main(argv=None):
    """Step 3 (and optionally step 4) from the command line.

      python -m warpstr_amd.wrapper --config cfg.yaml             every locus of a WarpSTR configuration (WarpSTR.py:33-76)
      python -m warpstr_amd.wrapper LOCUS_PATH SEQUENCE FLANK     one locus directory

    Under `python -m torch.distributed.run --nproc-per-node N -m warpstr_amd.wrapper ...` the work is sharded over the N
    GPUs of the node (whole loci from 8 loci per GPU on, else every locus's reads: warpstr_amd/loci.py)."""
    import argparse
    import time
    ap = argparse.ArgumentParser(prog='python -m warpstr_amd.wrapper', description=main.__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('locus_path', nargs='?')
    ap.add_argument('sequence', nargs='?')
    ap.add_argument('flank_length', type=int, nargs='?')
    ap.add_argument('--config', help="a WarpSTR YAML configuration (upstream's keys: output, loci, flank_length, threads, verbose, "
                                     'force_overwrite, pore_model_path, tr_calling_config, rescaling, genotyping_config, '
                                     'tr_region_calling, genotyping)')
    ap.add_argument('--segments-npz', dest='signals', help='.npz archive of normalised segments by read name (instead of the fast5 files)')
    ap.add_argument('--genotype', action='store_true', help='also run step 4 on the results')
    args = ap.parse_args(argv)
    shard = under_torchrun()
    from . import dist as wdist
    rank, world = (wdist.process_group() if shard else (0, 1))
    loader = _npz_loader(args.signals) if args.signals else None
    t_start = time.perf_counter()
    verbose, settings, own = 0, {}, None

    def duration(process, since):
        if verbose > 0 and rank == 0:  # (src/helpers.py:16-29, src/templates.py:3)
            t = time.perf_counter() - since
            print(f'  Duration for {process}: {int(t // 3600):02}h {int(t % 3600 // 60):02}m {int(t % 60):02}s')

    if args.config:
        from .config import load_config
        cfg = load_config(args.config)
        verbose, settings = cfg.verbose, cfg.genotyping_config.settings()
        loci = loci_from_config(cfg)
        genotype = args.genotype or cfg.genotyping
        if rank == 0:
            for line in cfg.notices():
                print('warpstr_amd: ' + line, file=sys.stderr)
            for locus in loci:  # (WarpSTR.py:42-45: the step's directories, and the sequence the locus was run with)
                if os.path.isdir(locus.path):
                    prepare_subdirs(locus.path, cfg.tr_region_calling, cfg.force_overwrite)
                    with open(os.path.join(locus.path, 'sequence.txt'), 'w') as f:
                        f.write(locus.sequence)
        if shard:
            import torch.distributed as tdist
            tdist.barrier()  # (the directories are as rank 0 left them before any rank writes into them)
        tables = [(None, None)] * len(loci)   # calling switched off: step 4 reads the overview.csv the earlier run left (WarpSTR.py:71-79)
        if cfg.tr_region_calling:
            tm = {}
            tables = main_wrapper_loci(loci, cfg.threads, caller_config=cfg.caller, rescaler_config=cfg.rescaler, signal_loader=loader,
                                       shard=shard, pore_model=cfg.pore_model(), timings=tm)
            if tm.get('partition') == 'loci':
                own = set(tm['loci_set_up'])   # every rank genotypes the loci it called
            duration('tr calling', t_start)
    else:
        if args.locus_path is None or args.sequence is None or args.flank_length is None:
            ap.error('either --config or LOCUS_PATH SEQUENCE FLANK_LENGTH')
        loci = [LocusPath(args.locus_path, args.sequence, args.flank_length)]
        genotype = args.genotype
        tables = [main_wrapper(loci[0], 1, signal_loader=loader, shard=shard)]
    t_gt = time.perf_counter()
    for i, (locus, pair) in enumerate(zip(loci, tables)):
        if (own is None and rank != 0) or (own is not None and i not in own):
            continue
        df_overview, df_collapsed = pair
        if df_overview is not None:
            called = int((np.asarray(df_overview['results']) >= 0).sum())
            print(f'{locus.name}: {called} reads called')
        if genotype:
            from .genotyper import run_genotyping_complex, run_genotyping_overview
            run_genotyping_overview(df_overview, locus.path, None, **settings)
            run_genotyping_complex(locus.path, df_collapsed, **settings)
    if genotype:
        duration('genotyping', t_gt)
    if verbose > 0 and rank == 0:
        duration('whole', t_start)
    if shard:
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized():
            tdist.barrier()
            tdist.destroy_process_group()


if __name__ == '__main__':
    main()

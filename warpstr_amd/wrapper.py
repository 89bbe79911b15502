"""main_wrapper: step 3 of the pipeline for one locus, on the GPU (src/caller/wrapper.py:17-54).

Reads <locus_path>/overview.csv and the flanks written by the squiggler step, calls every `saved` read
with the HIP caller, and writes the same outputs as the reference: overview.csv columns
(results, orig, dtw_cost1, dtw_cost2), predictions/sequences/*.fasta and, for loci with more than one
repeat unit, predictions/complexSTR_analysis/complex_repeat_units.csv.  Plots are not produced.
"""
import os
from typing import Callable, List, Optional

import numpy as np

from . import overview as ov
from .caller import CallerConfig, CallerWrapper, ReadSignal, RescalerConfig
from .signal_prep import process_raw
from .units import break_into_units, collapse_repeats

FAST5_SUBDIR, ANNOT_SUBDIR = 'fast5', 'annot'


def _fast5_loader(spike_removal: str) -> Callable[[str, int, int], np.ndarray]:
    try:
        import h5py  # noqa: F401
    except ImportError as e:  # pragma: no cover - h5py is absent in the build image
        raise RuntimeError('reading .fast5 needs h5py (and the VBZ HDF5 plugin); pass `signal_loader=` '
                           'to main_wrapper to supply normalised segments another way') from e

    def load(path: str, l_start_raw: int, r_end_raw: int) -> np.ndarray:
        import h5py
        with h5py.File(path, 'r') as h:
            rname = list(h['Raw']['Reads'].keys())[0]
            raw = np.asarray(h['Raw']['Reads'][rname]['Signal'])
        return process_raw(raw, (l_start_raw, r_end_raw), spike_removal)
    return load


def get_workload(df_overview, path: str, signal_loader: Callable[[str, int, int], np.ndarray]) -> List[ReadSignal]:
    """All `saved` rows of the overview, in overview order (src/caller/wrapper.py:44-54)."""
    work: List[ReadSignal] = []
    for row in df_overview.itertuples():
        if row.saved:
            fast5path = os.path.join(path, FAST5_SUBDIR, str(row.run_id), ANNOT_SUBDIR, row.Index + '.fast5')
            sig = signal_loader(fast5path, int(row.l_start_raw), int(row.r_end_raw))
            work.append(ReadSignal(row.Index, bool(row.reverse), np.asarray(sig, dtype=np.float64)))
    return work


def main_wrapper(locus_path: str, sequence: str, flank_length: int, threads: int = 1,
                 caller_config: Optional[CallerConfig] = None, rescaler_config: Optional[RescalerConfig] = None,
                 signal_loader: Optional[Callable[[str, int, int], np.ndarray]] = None, device: int = 0):
    caller_config = caller_config or CallerConfig()
    overview_path, df_overview = ov.load_overview(locus_path)
    loader = signal_loader or _fast5_loader(caller_config.spike_removal)
    workload = get_workload(df_overview, locus_path, loader)
    cw = CallerWrapper(sequence, ov.load_flanks(locus_path), flank_length, threads, caller_config, rescaler_config,
                       device=device)
    results = cw.run(workload)
    seq_results = [(r.seq, r.resc_seq) for r in results]
    cost_results = [(r.cost, r.resc_cost) for r in results]
    df_overview = ov.store_results(overview_path, df_overview, seq_results, cost_results, locus_path)
    df_collapsed = None
    units, repeat_units, offsets = break_into_units(sequence.upper())
    if len(units) > 1:
        collapsed = [collapse_repeats(s[1], repeat_units, offsets) for s in seq_results]
        df_collapsed = ov.store_collapsed(collapsed, units, repeat_units, [w.reverse for w in workload], locus_path)
    return df_overview, df_collapsed

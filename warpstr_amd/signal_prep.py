"""Raw squiggle -> normalised squiggle segment (host side of the loader).

Mirrors src/schemas/fast5.py:45-57,90-114: spike removal in the raw integer dtype, whole-read MAD
normalisation, then the slice [l_start_raw : r_end_raw + 1].
"""
import numpy as np

from .pore_model import normalize_signal_mad


def brute_remove(data: np.ndarray) -> np.ndarray:
    """Samples > 1000 or < 250 (at index > 2) are replaced, in order and in place of a copy, by the median
    of the 5-sample window around them (src/schemas/fast5.py:90-101)."""
    out = data.copy()
    for i in np.flatnonzero((data > 1000) | (data < 250)):
        if i > 2:
            out[i] = np.median(out[i - 2:i + 3])
    return out


def process_raw(raw: np.ndarray, position=None, spike_removal: str = 'Brute') -> np.ndarray:
    """Fast5.get_data_processed."""
    data = np.asarray(raw)
    if spike_removal == 'Brute':
        data = brute_remove(data)
    elif spike_removal in ('median3', 'median5'):
        from scipy.signal import medfilt
        data = medfilt(data, 3 if spike_removal == 'median3' else 5)
    norm = normalize_signal_mad(data)
    if position is not None:
        return norm[position[0]:position[1] + 1]
    return norm

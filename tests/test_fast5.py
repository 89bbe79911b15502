"""The .fast5 reader and the caller-only input path against the upstream test case's own data files
(tests/golden/real/: batch_0.fast5 + example.csv are copies of test/test_input/test_run1/fast5s/batch_0.fast5 and
test/test_caller_only/example.csv; real_aaat.npz holds what the upstream functions made of them).  CPU only."""
import os
import struct

import numpy as np
import pytest

from tests.helpers import GOLDEN, load_case
from warpstr_amd import fast5, overview as ov
from warpstr_amd.genotyper import call_alleles
from warpstr_amd.signal_prep import process_raw
from warpstr_amd.wrapper import get_raw_workload, prepare_caller_only

REAL = os.path.join(GOLDEN, 'real')
try:
    fast5._libs()
    HAVE_HDF5 = True
except fast5.Fast5Error:
    HAVE_HDF5 = False
needs_hdf5 = pytest.mark.skipif(not HAVE_HDF5, reason='no libhdf5/libzstd on this machine')


def svb_encode(vals):
    keys, body = bytearray((len(vals) + 3) // 4), bytearray()
    for i, v in enumerate(vals):
        nb = max(1, (int(v).bit_length() + 7) // 8)
        keys[i // 4] |= (nb - 1) << (2 * (i % 4))
        body += int(v).to_bytes(nb, 'little')
    return bytes(keys) + bytes(body)


@pytest.mark.parametrize('n', [0, 1, 3, 4, 5, 257])
def test_streamvbyte_and_vbz_chunk_round_trip(n):
    rng = np.random.default_rng(n)
    sig = rng.integers(-30000, 30000, size=n).astype(np.int16)
    if n > 4:
        sig[:4] = [0, 1, -1, 255]
    delta = np.diff(np.concatenate([[0], sig.astype(np.int64)]))  # |delta| reaches 60000: 1..3-byte codes
    zz = [(int(d) << 1) ^ (int(d) >> 63) for d in delta]
    chunk = struct.pack('<I', 2 * n) + svb_encode([z & 0xFFFFFFFF for z in zz])
    out = fast5.vbz_decode_chunk(chunk, 2, True, 0, 0)
    assert out.dtype == np.int16 and np.array_equal(out, sig)
    vals = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64)
    assert np.array_equal(fast5.streamvbyte_decode(np.frombuffer(svb_encode(vals), np.uint8), n), vals.astype(np.uint32))
    if n:
        with pytest.raises(fast5.Fast5Error):
            fast5.streamvbyte_decode(np.frombuffer(svb_encode(vals)[:-1], np.uint8), n)
    with pytest.raises(fast5.Fast5Error):
        fast5.vbz_decode_chunk(chunk, 2, True, 1, 0)  # 1-bit-key variant: refused, not mis-decoded


def test_c_decoder_equals_the_numpy_decoder(monkeypatch):
    """csrc/seam_helper.c's StreamVByte / zig-zag / running-sum loop (what fast5.py uses when the helper is built) against
    the NumPy decoder on random blocks: every code length, wrap-around of the int16 running sum, truncated blocks."""
    if fast5._vbz_c() is None:
        pytest.skip('warpstr_amd/_seam_helper.so is not built')
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 5, 64, 1000, 4097):
        for zigzag in (True, False):
            vals = rng.integers(0, 2 ** np.array([7, 8, 15, 16, 17, 24, 31, 32])[rng.integers(0, 8, size=n)], dtype=np.uint64)
            chunk = struct.pack('<I', 2 * n) + svb_encode([int(v) for v in vals])
            got = fast5.vbz_decode_chunk(chunk, 2, zigzag, 0, 0)
            with monkeypatch.context() as mp:
                mp.setattr(fast5, '_VBZ_C', None)
                want = fast5.vbz_decode_chunk(chunk, 2, zigzag, 0, 0)
            assert got.dtype == np.int16 and np.array_equal(got, want), (n, zigzag)
            if n:
                with pytest.raises(fast5.Fast5Error):
                    fast5.vbz_decode_chunk(chunk[:-1], 2, zigzag, 0, 0)
                with pytest.raises(fast5.Fast5Error):
                    fast5.vbz_decode_chunk(chunk[:4 + (n + 3) // 4 - 1], 2, zigzag, 0, 0)


@needs_hdf5
def test_reader_on_upstream_multi_read_file():
    z = load_case('real_aaat')
    with fast5.Fast5File(os.path.join(REAL, 'batch_0.fast5')) as f:
        ids = f.read_ids()
        assert sorted(ids) == sorted(str(n) for n in z['names']) and len(ids) == 10
        for i, name in enumerate(z['names']):
            raw = f.raw_signal(str(name))
            lo, hi, n = (int(v) for v in z['raw_span'][i])
            assert raw.dtype == np.int16 and raw.size == n
            # spike removal + whole-read normalisation + slice: bit-identical to what the upstream functions produced
            assert np.array_equal(process_raw(raw, (lo, hi)), z[f'r{i}_signal'])
        with pytest.raises(fast5.Fast5Error):
            f.raw_signal('no-such-read')
    with pytest.raises(fast5.Fast5Error):
        fast5.read_raw_signal(os.path.join(REAL, 'example.csv'))


@needs_hdf5
def test_prepare_caller_only_and_raw_workload(tmp_path):
    # the csv names the fast5 relative to the upstream checkout
    d = tmp_path / 'test' / 'test_input' / 'test_run1' / 'fast5s'
    d.mkdir(parents=True)
    os.symlink(os.path.join(REAL, 'batch_0.fast5'), d / 'batch_0.fast5')
    loci = prepare_caller_only(os.path.join(REAL, 'example.csv'), str(tmp_path / 'out'), base_dir=str(tmp_path))
    assert list(loci) == ['Human_STR_1108232']
    _, df = ov.load_overview(loci['Human_STR_1108232'])
    assert list(df.columns) == ['fast5_path', 'locus', 'reverse', 'l_start_raw', 'r_end_raw', 'run_id', 'saved']
    assert set(df['run_id']) == {'run_0'} and set(df['saved']) == {1}
    names, revs, raws, pos = get_raw_workload(df, loci['Human_STR_1108232'])
    z = load_case('real_aaat')
    assert names == [str(n) for n in z['names']] and revs == [bool(r) for r in z['reverse']]
    assert [p[0] for p in pos] == [int(v) for v in z['raw_span'][:, 0]]
    assert [len(r) for r in raws] == [int(v) for v in z['raw_span'][:, 2]]
    bad = tmp_path / 'bad.csv'
    bad.write_text('fast5_path,locus,read_name\nx,y,z\n')
    with pytest.raises(ValueError):
        prepare_caller_only(str(bad), str(tmp_path / 'out2'))
    with pytest.raises(FileNotFoundError):
        prepare_caller_only(os.path.join(REAL, 'example.csv'), str(tmp_path / 'out3'), base_dir=str(tmp_path / 'nowhere'))


def test_upstream_known_answer_from_golden_lengths():
    """README.md section 2: 'Allele lengths as given by WarpSTR: (44, 40)'.  The per-read lengths recorded from the
    upstream caller on these reads genotype to exactly that."""
    z = load_case('real_aaat')
    lens = [len(str(z[f'r{i}_seq'][1])) for i in range(int(z['n_reads']))]
    assert sorted(set(lens)) == [40, 44]
    gt = call_alleles(lens, random_state=0)
    assert gt.heterozygous and sorted(gt.alleles, reverse=True) == [44, 40]

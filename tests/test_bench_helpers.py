"""The host-side arithmetic of bench.py (no GPU): union of fill intervals, the oracle check of a bench line, the strict
lookup of profiled counters, the strong-scaling partition."""
import json
import os
import types

import numpy as np
import pytest

import bench
from warpstr_amd import _lib
from warpstr_amd.dist import shard_reads


def test_union_of_overlapping_launch_intervals():
    b = np.array([0.0, 1.0, 5.0, 5.5, 20.0])
    e = np.array([2.0, 3.0, 6.0, 5.8, 21.0])
    assert bench.union_ms(b, e) == pytest.approx(3.0 + 1.0 + 1.0)
    assert bench.union_ms(np.array([]), np.array([])) == 0.0
    assert bench.union_ms(np.array([3.0, 0.0]), np.array([4.0, 1.0])) == pytest.approx(2.0)   # unsorted input
    assert bench.union_ms(b, e) <= float((e - b).sum())


def test_verify_counts_mismatching_reads():
    rec = np.zeros(4, dtype=_lib.RESULT_DTYPE)
    rec['len1'], rec['len2'] = [10, 11, 12, 0], [9, 11, 12, 0]
    rec['cost1'], rec['cost2'] = [0.5, 0.25, 0.125, np.nan], [0.4, 0.2, 0.1, np.nan]
    rec['status'] = [0, 0, 0, 3]
    o = lambda st, l1, l2, c1, c2: types.SimpleNamespace(status=st, len1=l1, len2=l2, cost1=c1, cost2=c2)
    good = [o(0, 10, 9, 0.5, 0.4), o(0, 11, 11, 0.25 * (1 + 5e-6), 0.2), o(0, 12, 12, 0.125, 0.1), o(3, 0, 0, float('nan'), float('nan'))]
    assert bench.verify(rec, good)['mismatches'] == 0
    bad = list(good)
    bad[1] = o(0, 11, 12, 0.25, 0.2)          # allele length differs
    bad[2] = o(0, 12, 12, 0.125, 0.1002)      # cost outside 1e-5
    v = bench.verify(rec, bad)
    assert v['mismatches'] == 2 and v['first_mismatches'] == [1, 2] and v['reads'] == 4
    assert bench.verify(rec, [o(1, 0, 0, 0, 0)])['mismatches'] == 1   # status differs


def test_profiled_counters_must_belong_to_the_kernel(monkeypatch, tmp_path):
    """bench.py quotes PMC counters only for the kernel NAME they were taken from and only while the kernel SOURCES are the
    ones they were taken on (kernel_source_hash in every entry of profiles/fill_pmc.json)."""
    table = json.load(open(os.path.join(bench.ROOT, 'profiles', 'fill_pmc.json')))
    kernel = sorted(table)[0]
    monkeypatch.delenv('WARPSTR_BENCH_PROFILING', raising=False)
    with pytest.raises(SystemExit):
        bench.fill_profile('dtw_fill_fast<9, 9, 9, 9, false>')
    # an entry measured on other sources is not quoted
    real_hash = bench.kernel_source_hash
    monkeypatch.setattr(bench, 'kernel_source_hash', lambda: 'somethingelse')
    assert 'stale' in bench.fill_profile(kernel)
    monkeypatch.setattr(bench, 'kernel_source_hash', lambda: table[kernel].get('kernel_source_hash'))
    prof = bench.fill_profile(kernel)
    assert prof['source'].startswith('profiles/') and prof['valu_insts_per_wave_row'] > 8
    r = bench.valu_roofline(prof, 5.0, 2e8, kernel)
    assert r['counters_from'] == prof['source'] and 0 < r['frac'] < r['frac_at_observed_clock'] < 1.2
    monkeypatch.setattr(bench, 'kernel_source_hash', real_hash)
    monkeypatch.setenv('WARPSTR_BENCH_PROFILING', '1')
    assert bench.fill_profile('dtw_fill_fast<9, 9, 9, 9, false>') is None


def test_committed_counters_are_those_of_the_committed_kernels():
    """The headline kernel's entry of profiles/fill_pmc.json was measured on the kernel sources in this tree."""
    table = json.load(open(os.path.join(bench.ROOT, 'profiles', 'fill_pmc.json')))
    entry = table['dtw_fill_fast<4, 1, 2, 2, true, 0>']
    assert entry.get('kernel_source_hash') == bench.kernel_source_hash(), \
        'kernel sources changed since the PMC passes: re-run scripts/profile_round.sh + scripts/summarize_profiles.py'


def test_strong_scaling_partition_covers_the_workload_once():
    for world in (1, 2, 4, 8):
        shards = shard_reads(np.full(100000, 2000, np.int64), world)
        allr = np.concatenate(shards)
        assert len(allr) == 100000 and len(np.unique(allr)) == 100000
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 1 and max(sizes) == (100000 + world - 1) // world

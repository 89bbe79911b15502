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


def test_bare_gpus_n_without_the_gpus_ends_with_one_sentence():
    """`python bench.py --gpus 8` with no launcher and fewer than 8 GPUs visible (here: none): no rank is started, exit code 2 and
    one sentence on stderr -- the driver's first scaling run must not die in a stack trace."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8'], capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip('eight GPUs here')
    assert out.returncode == 2 and out.stdout.strip() == ''
    assert 'needs 8 GPUs on this node' in out.stderr and 'Traceback' not in out.stderr


def test_launcher_relays_the_line_and_the_exit_code(tmp_path, monkeypatch):
    """bench.launch_ranks: the child torch.distributed.run's rank-0 line goes to stdout alone, everything else to stderr, and
    the job's exit code is returned (here the `ranks` are a stand-in script: two gloo ranks, no GPU)."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    script = tmp_path / 'fake_bench.py'
    script.write_text(
        "import os, sys, json\n"
        "import torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "dist.barrier()\n"
        "if r == 0:\n"
        "    print('banner of some library')\n"
        "    print(json.dumps({'metric': 'reads/s', 'world': w, 'argv': sys.argv[1:]}))\n"
        "dist.destroy_process_group()\n"
        "sys.exit(5 if '--fail' in sys.argv else 0)\n")
    monkeypatch.setenv('WARPSTR_BENCH_BACKEND', 'gloo')
    monkeypatch.setattr(bench, '__file__', str(script))
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import bench; bench.__file__ = %r; "
            "sys.exit(bench.launch_ranks(2, sys.argv[1:]))" % (root, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['WARPSTR_BENCH_BACKEND'] = 'gloo'
    ok = subprocess.run([sys.executable, '-c', code, '--steps', '3'], capture_output=True, text=True, timeout=300, env=env)
    assert ok.returncode == 0, ok.stderr[-2000:]
    lines = [l for l in ok.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {'metric': 'reads/s', 'world': 2, 'argv': ['--steps', '3']}
    assert 'banner of some library' in ok.stderr
    bad = subprocess.run([sys.executable, '-c', code, '--fail'], capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode != 0

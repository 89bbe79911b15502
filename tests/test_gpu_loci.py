"""Several loci through ONE handle (warpstr_amd/loci.py: main_wrapper_loci; `python -m warpstr_amd.wrapper --config`): mixed-locus
batches, pipelined, against one main_wrapper call per locus -- every output file byte for byte.  In process (raw int16 reads
through the GPU loader) and from the command line (a WarpSTR YAML; one rank and two gloo ranks on the one card)."""
import filecmp
import os
import shutil
import socket
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from warpstr_amd import overview as ov, synth
from warpstr_amd.wrapper import LocusPath, main_wrapper, main_wrapper_loci

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUTPUTS = ['overview.csv', 'predictions/sequences/all.fasta', 'predictions/sequences/sequences_template.fasta',
           'predictions/sequences/sequences_reverse.fasta', 'summaries/state_similarity.csv']
COMPLEX = 'predictions/complexSTR_analysis/complex_repeat_units.csv'
# pattern, flank, reads, T: one to four slots, two to four candidates, a locus without saved reads, one with a single read
LOCI = [('(AGC)', 16, 33, (900, 1600)), ('(AGC)AACAGCCGCCAC(CGC)', 20, 40, (1400, 2200)), ('(AAAT)', 110, 21, (2300, 3100)),
        ('((CAGG){CAGM})(CAGA)(CA)', 40, 27, (1500, 3000)), ('(GGCCCC)', 30, 0, (900, 1000)), ('(NGC)', 24, 16, (1000, 1800)),
        ('(CAG)CAACAG(CCG)', 70, 1, (2000, 2400)), ('(CCTG)(TCTG)', 110, 18, (2400, 3300))]


def _free_port():
    from tests.helpers import free_port
    return free_port()


def _make(root, tag):
    """Locus directories (overview + flanks) under root/tag; returns (loci, raw reads by name, normalised segments by name)."""
    loci, raws, segs = [], {}, {}
    for li, (pattern, fl, n, T) in enumerate(LOCI):
        locus = synth.make_locus(pattern, fl, 300 + li)
        sigs, revs, _ = synth.batch(locus, n, T, 400 + li, lo=3, hi=10)
        loc = os.path.join(root, tag, f'locus{li}')
        ov.store_flanks(loc, [locus.left_t, locus.right_t, locus.left_r, locus.right_r])
        names = [f'L{li}r{i:03d}' for i in range(n + 2)]
        rng = np.random.default_rng(500 + li)
        lo, hi = [], []
        for nm, s in zip(names, sigs):  # the raw read: the segment with some signal on both sides, as DAC values
            pre, post = int(rng.integers(50, 400)), int(rng.integers(50, 400))
            whole = np.concatenate([rng.normal(0, 1, pre), s, rng.normal(0, 1, post)])
            raws[nm] = np.clip(np.round(whole * 70.0 + 500.0), 0, 2047).astype(np.int16)
            segs[nm] = s
            lo.append(pre)
            hi.append(pre + len(s) - 1)
        pd.DataFrame({'read_name': names, 'run_id': 0, 'reverse': list(revs) + [False, True], 'saved': [1] * n + [0, 0],
                      'l_start_raw': lo + [10, 10], 'r_end_raw': hi + [90, 90]}).to_csv(os.path.join(loc, 'overview.csv'), index=False)
        loci.append(LocusPath(loc, pattern, fl))
    return loci, raws, segs


def _same(a, b):
    for la, lb in zip(a, b):
        rels = OUTPUTS + ([COMPLEX] if os.path.exists(os.path.join(la.path, COMPLEX)) else [])
        for rel in rels:
            assert filecmp.cmp(os.path.join(la.path, rel), os.path.join(lb.path, rel), shallow=False), (la.path, rel)


def test_eight_loci_in_one_handle_equal_eight_main_wrapper_calls(tmp_path):
    """From raw int16 reads (the default path: loader kernels + caller, both called sequences packed on the device), cut into
    batches of at most 30 reads so that several are in flight and loci straddle them."""
    one, raws, _ = _make(str(tmp_path), 'one')
    many, _, _ = _make(str(tmp_path), 'many')
    reader = lambda path: raws[os.path.basename(path)[:-len('.fast5')]]
    for locus in one:
        main_wrapper(locus, 1, raw_reader=reader)
    tm = {}
    tables = main_wrapper_loci(many, 1, raw_reader=reader, batch_reads=30, timings=tm, quiet=True)
    _same(one, many)
    assert tm['n_loci'] == 8 and tm['n_reads'] == sum(n for _, _, n, _ in LOCI) and len(tm['kernels']) >= 4
    assert os.path.exists(os.path.join(many[1].path, COMPLEX)) and os.path.exists(os.path.join(many[3].path, COMPLEX))
    for (df, _), (_, _, n, _) in zip(tables, LOCI):
        assert int((df['results'] >= 0).sum()) == n
    # the same loci again, everything in one batch: nothing depends on where the list was cut
    again, _, _ = _make(str(tmp_path), 'again')
    main_wrapper_loci(again, 1, raw_reader=reader, quiet=True)
    _same(one, again)


def _cli(args, ranks, extra_env=None, module='warpstr_amd.wrapper'):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    env.update(extra_env or {})
    tail = ['-m', module] + args
    if ranks == 1:
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={ranks}', '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port())] + tail
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout


def test_config_driven_run_on_one_and_two_ranks(tmp_path):
    """`python -m warpstr_amd.wrapper --config cfg.yaml` (upstream: `python WarpSTR.py cfg.yaml` with tr_region_calling only):
    one rank and two ranks (gloo collectives, both on the one card) write what eight per-locus command lines write."""
    one, _, segs = _make(str(tmp_path), 'one')
    npz = str(tmp_path / 'segments.npz')
    np.savez(npz, **segs)
    for locus in one:
        _cli([locus.path, locus.sequence, str(locus.flank_length), '--segments-npz', npz], 1)
    for tag, ranks, env in (('cfg1', 1, None), ('cfg2', 2, {'WARPSTR_DIST_BACKEND': 'gloo'})):
        shutil.copytree(os.path.join(tmp_path, 'one'), os.path.join(tmp_path, tag),
                        ignore=shutil.ignore_patterns('predictions', 'summaries'))
        for li in range(len(LOCI)):  # (the copied overviews carry the first run's result columns: start from fresh ones)
            df = pd.read_csv(os.path.join(tmp_path, tag, f'locus{li}', 'overview.csv'))
            df.drop(columns=['results', 'orig', 'dtw_cost1', 'dtw_cost2']).to_csv(os.path.join(tmp_path, tag, f'locus{li}', 'overview.csv'), index=False)
        cfg = tmp_path / f'{tag}.yaml'
        cfg.write_text(f'output: {tmp_path / tag}\nthreads: 2\ntr_region_calling: True\ngenotyping: False\nloci:\n' + ''.join(
            f'  - name: locus{li}\n    coord: chr1:1-2\n    sequence: {p}\n    flank_length: {fl}\n' for li, (p, fl, _, _) in enumerate(LOCI)))
        if ranks == 1:  # upstream's command line: `python WarpSTR.py cfg.yaml` -> `python -m warpstr_amd cfg.yaml`
            out = _cli([str(cfg), '--segments-npz', npz], 1, env, module='warpstr_amd')
        else:
            out = _cli(['--config', str(cfg), '--segments-npz', npz], ranks, env)
        for li, (_, _, n, _) in enumerate(LOCI):
            assert f'locus{li}: {n} reads called' in out
        _same(one, [LocusPath(os.path.join(tmp_path, tag, f'locus{li}'), p, fl) for li, (p, fl, _, _) in enumerate(LOCI)])


def test_configuration_keys_reach_the_kernels_and_the_genotyper(tmp_path, capsys):
    """`python -m warpstr_amd cfg.yaml` with a perturbed `pore_model_path`, non-default `genotyping_config` and `force_overwrite`
    (tests/golden/cfg_keys.*: upstream run with the same table and settings): every read's called lengths and costs as upstream's
    caller gave them under THAT table, alleles.csv as upstream's genotyper wrote it under THOSE settings, last run's files gone."""
    import json

    import yaml

    from tests.helpers import GOLDEN, load_case, write_perturbed_pore_model
    from warpstr_amd.wrapper import main
    fix = json.load(open(os.path.join(GOLDEN, 'cfg_keys.json')))
    z = load_case('cfg_keys')
    n = int(z['n_reads'])
    loc = tmp_path / 'out' / 'L'
    ov.store_flanks(str(loc), [str(x) for x in z['flanks']])
    names = [f'read{i:03d}' for i in range(n)]
    pd.DataFrame({'read_name': names, 'run_id': 0, 'reverse': z['reverse'].astype(bool), 'saved': 1, 'l_start_raw': 0,
                  'r_end_raw': [len(z[f'r{i}_signal']) - 1 for i in range(n)]}).to_csv(loc / 'overview.csv', index=False)
    os.makedirs(loc / 'predictions' / 'sequences')
    (loc / 'predictions' / 'sequences' / 'stale.fasta').write_text('>old\n')
    npz = str(tmp_path / 'segments.npz')
    np.savez(npz, **{nm: z[f'r{i}_signal'] for i, nm in enumerate(names)})
    cfg = {'output': str(tmp_path / 'out'), 'threads': 2, 'tr_region_calling': True, 'genotyping': True, 'force_overwrite': True,
           'flank_length': fix['flank_length'], 'pore_model_path': write_perturbed_pore_model(str(tmp_path / 'perturbed.model')),
           'tr_calling_config': {'visualize_alignment': False, 'visualize_phase': False, 'visualize_strand': False, 'visualize_cost': False},
           'genotyping_config': fix['genotyping_config'], 'loci': [{'name': 'L', 'coord': 'chr1:1-2', 'sequence': fix['pattern']}]}
    path = str(tmp_path / 'cfg.yaml')
    with open(path, 'w') as f:
        yaml.safe_dump(cfg, f)
    case = fix['cases'][0]
    np.random.seed(case['seed'])
    main(['--config', path, '--segments-npz', npz])
    df = pd.read_csv(loc / 'overview.csv')
    for i in range(n):
        assert (int(df['orig'][i]), int(df['results'][i])) == (len(str(z[f'r{i}_seq'][0])), len(str(z[f'r{i}_seq'][1]))), i
        assert abs(df['dtw_cost1'][i] - z[f'r{i}_cost'][0]) <= 1e-9 * abs(z[f'r{i}_cost'][0])
        assert abs(df['dtw_cost2'][i] - z[f'r{i}_cost'][1]) <= 1e-9 * abs(z[f'r{i}_cost'][1])
    assert [int(v) for v in df['results']] == case['results']
    assert open(loc / 'predictions' / 'alleles.csv').read() == case['alleles_csv']
    assert case['stdout'].strip() in capsys.readouterr().out
    assert not (loc / 'predictions' / 'sequences' / 'stale.fasta').exists()
    # the same reads under the default table are called differently: the key is not decoration
    cfg2 = dict(cfg, pore_model_path='example/deps/template_median68pA.model', genotyping=False)
    with open(path, 'w') as f:
        yaml.safe_dump(cfg2, f)
    main(['--config', path, '--segments-npz', npz])
    df2 = pd.read_csv(loc / 'overview.csv')
    assert not np.allclose(df2['dtw_cost2'], df['dtw_cost2'])


MANY = [('(AGC)', 16, (900, 1400)), ('(AAAT)', 40, (1200, 1800)), ('(AGC)AACAGCCGCCAC(CGC)', 20, (1300, 1900)), ('(GGCCCC)', 30, (1000, 1500)),
        ('(CTG)CTA(CTG)', 24, (1200, 1700))]


def _make_many(root, tag, n_loci):
    """n_loci small loci (1-6 reads, one without saved reads) + the normalised segments by read name."""
    loci, segs = [], {}
    for li in range(n_loci):
        pattern, fl, T = MANY[li % len(MANY)]
        n = 0 if li == 7 else 1 + (li * 5) % 6
        locus = synth.make_locus(pattern, fl, 700 + li)
        sigs, revs, _ = synth.batch(locus, n, T, 800 + li, lo=3, hi=10)
        loc = os.path.join(root, tag, f'locus{li:02d}')
        ov.store_flanks(loc, [locus.left_t, locus.right_t, locus.left_r, locus.right_r])
        names = [f'M{li:02d}r{i}' for i in range(n + 1)]
        pd.DataFrame({'read_name': names, 'run_id': 0, 'reverse': list(revs) + [False], 'saved': [1] * n + [0], 'l_start_raw': 0,
                      'r_end_raw': [len(s) - 1 for s in sigs] + [50]}).to_csv(os.path.join(loc, 'overview.csv'), index=False)
        segs.update(dict(zip(names, sigs)))
        loci.append(LocusPath(loc, pattern, fl, f'locus{li:02d}'))
    return loci, segs


def _many_cfg(tmp_path, tag, n_loci):
    cfg = tmp_path / f'{tag}.yaml'
    cfg.write_text(f'output: {tmp_path / tag}\nthreads: 2\ntr_region_calling: True\ngenotyping: False\nloci:\n' + ''.join(
        f'  - name: locus{li:02d}\n    coord: chr1:1-2\n    sequence: {MANY[li % len(MANY)][0]}\n    flank_length: {MANY[li % len(MANY)][1]}\n'
        for li in range(n_loci)))
    return str(cfg)


def test_forty_loci_partitioned_over_four_ranks_on_the_one_card(tmp_path):
    """The locus partition of a many-loci run (>= 8 loci per rank) on real kernels: four fresh ranks on the one card (gloo for the
    collectives), every rank setting up, calling and writing only its own loci -- the files of one rank's run, byte for byte."""
    n_loci = 40
    one, segs = _make_many(str(tmp_path), 'one', n_loci)
    _make_many(str(tmp_path), 'four', n_loci)
    npz = str(tmp_path / 'segments.npz')
    np.savez(npz, **segs)
    out1 = _cli(['--config', _many_cfg(tmp_path, 'one', n_loci), '--segments-npz', npz], 1)
    out4 = _cli(['--config', _many_cfg(tmp_path, 'four', n_loci), '--segments-npz', npz], 4, {'WARPSTR_DIST_BACKEND': 'gloo'})
    for li in range(n_loci):
        n = 0 if li == 7 else 1 + (li * 5) % 6
        assert f'locus{li:02d}: {n} reads called' in out1 and f'locus{li:02d}: {n} reads called' in out4
    _same(one, [LocusPath(os.path.join(tmp_path, 'four', f'locus{li:02d}'), *MANY[li % len(MANY)][:2]) for li in range(n_loci)])


@pytest.mark.parametrize('partition', ['loci', 'reads'])
def test_many_loci_through_a_one_rank_rccl_group(tmp_path, partition):
    """The collectives of a sharded many-loci run over the nccl backend (RCCL), as far as one GPU can show them: a one-rank group
    (WARPSTR_DIST_SELF_GATHER=1), partition by locus (counts + barrier) and by read (records and sequences all-gathered) -- the
    files of a plain run."""
    n_loci = 24
    one, segs = _make_many(str(tmp_path), 'one', n_loci)
    _make_many(str(tmp_path), 'rccl', n_loci)
    npz = str(tmp_path / 'segments.npz')
    np.savez(npz, **segs)
    _cli(['--config', _many_cfg(tmp_path, 'one', n_loci), '--segments-npz', npz], 1)
    script = (f"import os, sys, json; sys.path.insert(0, {ROOT!r})\n"
              "import numpy as np\n"
              "from warpstr_amd.wrapper import LocusPath, main_wrapper_loci, _npz_loader\n"
              f"loci = [LocusPath(os.path.join({str(tmp_path / 'rccl')!r}, 'locus%02d' % li), *{MANY!r}[li % {len(MANY)}][:2]) for li in range({n_loci})]\n"
              f"tm = {{}}\nmain_wrapper_loci(loci, 2, signal_loader=_npz_loader({npz!r}), shard=True, partition={partition!r}, quiet=True, timings=tm)\n"
              "import torch.distributed as d\nprint(json.dumps({'backend': d.get_backend(), 'partition': tm['partition'], 'gather_s': tm['gather_s']}))\nd.destroy_process_group()\n")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', WARPSTR_DIST_SELF_GATHER='1', WARPSTR_DIST_BACKEND='nccl', MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    info = json.loads(out.stdout.strip().splitlines()[-1])
    assert info['backend'] == 'nccl' and info['partition'] == partition
    _same(one, [LocusPath(os.path.join(tmp_path, 'rccl', f'locus{li:02d}'), *MANY[li % len(MANY)][:2]) for li in range(n_loci)])


@pytest.mark.parametrize('mode', ['arenas, zstd and VBZ decoded on the GPU', 'arenas, VBZ decoded on the GPU', 'arenas', 'shared staging'])
def test_fast5_reads_decoded_by_reader_processes_into_the_upload_buffers(tmp_path, monkeypatch, mode):
    """The path real input takes, on the GPU: 70 loci whose reads are the upstream test file's ten VBZ reads (caller-only layout),
    read in this process against three reader processes that decode straight into page-locked memory both sides map -- arenas of
    their own, handed out a batch ahead (_readers.decode_arena; or the chunks' zstd frames as they lie in the file, the rest on
    the device: wsx_zstd_decode + wsx_vbz_decode; or as far as the StreamVByte blocks inside those frames, _readers.pack_arena), or the staging ring of caller.SharedStaging (lengths
    first, then every read to its place) -- the same files, and the reads were uploaded from there."""
    if mode == 'shared staging':
        monkeypatch.setenv('WARPSTR_NO_READER_ARENAS', '1')
    if mode == 'arenas':
        monkeypatch.setenv('WARPSTR_NO_GPU_VBZ', '1')
    if mode == 'arenas, VBZ decoded on the GPU':
        monkeypatch.setenv('WARPSTR_NO_GPU_ZSTD', '1')
    monkeypatch.setattr('warpstr_amd.loci.SHARED_BATCH_READS', 24)   # (several batches: every arena region is used again)
    from tests.helpers import GOLDEN
    from warpstr_amd import fast5
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    src = os.path.join(GOLDEN, 'real', 'batch_0.fast5')
    ids = fast5.Fast5File(src).read_ids()[:10]

    def make(root):
        loci = []
        for li in range(70):
            pattern, fl, _ = MANY[li % len(MANY)]
            locus = synth.make_locus(pattern, fl, 900 + li)
            loc = os.path.join(root, f'locus{li}')
            ov.store_flanks(loc, [locus.left_t, locus.right_t, locus.left_r, locus.right_r])
            rows = [ids[(li + k) % 10] for k in range(1 + li % 3)]
            pd.DataFrame({'read_name': rows, 'run_id': 'run_0', 'reverse': [bool((li + k) & 1) for k in range(len(rows))], 'saved': 1,
                          'l_start_raw': 5000 + 10 * li, 'r_end_raw': 6500 + 10 * li, 'fast5_path': src}).to_csv(os.path.join(loc, 'overview.csv'), index=False)
            loci.append(LocusPath(loc, pattern, fl))
        return loci
    a, b = make(str(tmp_path / 'a')), make(str(tmp_path / 'b'))
    tm_a, tm_b = {}, {}
    main_wrapper_loci(a, 3, quiet=True, timings=tm_a)
    main_wrapper_loci(b, 1, quiet=True, timings=tm_b)
    assert tm_a['reader_processes'] == 3 and tm_a.get('raw_bytes', 0) > 0 and not tm_a.get('shared_staging_refused')
    assert tm_a['reader_mode'] == mode
    if 'VBZ' in mode:
        assert 0 < tm_a['uploaded_bytes'] < 0.75 * tm_a['raw_bytes']
    assert tm_b['reader_processes'] == 0
    _same(a, b)


def test_reads_whose_datasets_are_several_vbz_chunks(tmp_path, monkeypatch):
    """Reads stored as several VBZ chunks each (tests/helpers.write_vbz_fast5: the last chunk codes a whole chunk of which the
    dataset holds a part, one chunk has the filter skipped): several blocks per read through wsx_vbz_decode -- the samples of a
    read's later blocks land behind its earlier ones -- against the same run with the readers decoding themselves."""
    from tests.helpers import write_vbz_fast5
    from warpstr_amd import fast5
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    # the upstream test file's ten reads (59-170 k samples), stored again in chunks of 20 480 samples: 3-9 blocks per read
    from tests.helpers import GOLDEN
    with fast5.Fast5File(os.path.join(GOLDEN, 'real', 'batch_0.fast5')) as f:
        reads = {rid: f.raw_signal(rid) for rid in f.read_ids()[:10]}
    src = write_vbz_fast5(str(tmp_path / 'chunks.fast5'), reads, 20480, skip_filter_on=(2,))
    ids = list(reads)

    def make(root):
        loci = []
        for li in range(70):
            pattern, fl, _ = MANY[li % len(MANY)]
            locus = synth.make_locus(pattern, fl, 900 + li)
            loc = os.path.join(root, f'locus{li}')
            ov.store_flanks(loc, [locus.left_t, locus.right_t, locus.left_r, locus.right_r])
            rows = [ids[(li + k) % 10] for k in range(1 + li % 3)]
            pd.DataFrame({'read_name': rows, 'run_id': 'run_0', 'reverse': [bool((li + k) & 1) for k in range(len(rows))], 'saved': 1,
                          'l_start_raw': 5000 + 10 * li, 'r_end_raw': 6500 + 10 * li, 'fast5_path': src}).to_csv(os.path.join(loc, 'overview.csv'), index=False)
            loci.append(LocusPath(loc, pattern, fl))
        return loci
    monkeypatch.setattr('warpstr_amd.loci.SHARED_BATCH_READS', 24)
    a, b, c = make(str(tmp_path / 'a')), make(str(tmp_path / 'b')), make(str(tmp_path / 'c'))
    tm_a, tm_b, tm_c = {}, {}, {}
    main_wrapper_loci(a, 3, quiet=True, timings=tm_a)
    main_wrapper_loci(c, 1, quiet=True, timings=tm_c)
    monkeypatch.setenv('WARPSTR_NO_GPU_VBZ', '1')
    main_wrapper_loci(b, 3, quiet=True, timings=tm_b)
    assert tm_a['reader_mode'] == 'arenas, zstd and VBZ decoded on the GPU' and tm_b['reader_mode'] == 'arenas'
    assert tm_c['reader_mode'] == 'arenas, zstd and VBZ decoded on the GPU, filled in this process'
    assert tm_a['uploaded_bytes'] < tm_a['raw_bytes'] == tm_b['raw_bytes'] == tm_c['raw_bytes']
    _same(a, b)
    _same(c, b)


def test_a_block_that_changed_on_its_way_to_the_device_fails_its_batch(tmp_path):
    """The readers check every StreamVByte block before it is uploaded; should one still reach the device with keys that ask for
    more bytes than it has (memory that changed in between), the device decoder flags it and collect() raises -- the batch is
    not called on samples that were never decoded.  Built by hand: an arena file with a good and a tampered block."""
    import tempfile
    import torch
    from oracle import vbz
    from warpstr_amd import _lib
    from warpstr_amd.loci import HipEngine
    if not os.path.isdir('/dev/shm'):
        pytest.skip('no /dev/shm')
    locus = synth.make_locus('(AGC)', 16, 3)
    engine = HipEngine([locus.template, locus.reverse], [16, 16], None, None, 0)
    rng = np.random.default_rng(4)
    raws = [np.cumsum(rng.integers(-30, 31, size=6000)).astype(np.int16) + 500 for _ in range(2)]
    blocks = [vbz.svb_encode(vbz.values_from_samples(r, True)) for r in raws]
    blocks[1] = blocks[1].copy()
    blocks[1][700:1500] = 0xFF                     # the second block's keys now ask for far more bytes than follow them
    fd, path = tempfile.mkstemp(prefix='warpstr_test_arena_', dir='/dev/shm')
    try:
        offs, at = [], 0
        for b in blocks:
            at = (at + 15) & ~15
            offs.append(at)
            at += len(b)
        cap = 1 << 20
        os.ftruncate(fd, cap)
        with os.fdopen(fd, 'r+b') as fh:
            for o, b in zip(offs, blocks):
                fh.seek(o)
                fh.write(b.tobytes())
        table = np.array([[0, _lib.VBZ_SVB_ZIGZAG, offs[0], len(blocks[0]), 6000, 6000, 0],
                          [1, _lib.VBZ_SVB_ZIGZAG, offs[1], len(blocks[1]), 6000, 6000, 0]], np.int64)
        part = (path, cap, 0, at, [6000, 6000], table.tobytes())
        lo, hi, aut = np.array([1000, 1000]), np.array([2999, 2999]), np.array([0, 1], np.int32)
        ticket = engine.submit_vbz_parts(0, [part], lo, hi, aut)
        with pytest.raises(RuntimeError, match='the device decoders flagged a chunk'):
            engine.collect(ticket)
        # the same arena with the good block twice: called as ever
        good = np.array([[0, _lib.VBZ_SVB_ZIGZAG, offs[0], len(blocks[0]), 6000, 6000, 0],
                         [1, _lib.VBZ_SVB_ZIGZAG, offs[0], len(blocks[0]), 6000, 6000, 0]], np.int64)
        ticket = engine.submit_vbz_parts(1, [(path, cap, 0, at, [6000, 6000], good.tobytes())], lo, hi, aut)
        rec = engine.collect(ticket)[0]
        assert len(rec) == 2
        # ... and a chunk handed over as its zstd frame (kind 3) whose frame is cut short: wsx_zstd_decode flags it
        from tests.test_zstd_oracle import compress
        frame = compress(blocks[0].tobytes(), 1)
        with open(path, 'r+b') as fh:
            fh.seek(offs[0])
            fh.write(frame[:-9] + bytes(9))
        framed = np.array([[0, 3, offs[0], len(frame), 6000, 6000, len(blocks[0])], [1, _lib.VBZ_SVB_ZIGZAG, offs[1], len(blocks[0]), 6000, 6000, 0]], np.int64)
        with open(path, 'r+b') as fh:
            fh.seek(offs[1])
            fh.write(blocks[0].tobytes())
        ticket = engine.submit_vbz_parts(2, [(path, cap, 0, at, [6000, 6000], framed.tobytes())], lo, hi, aut)
        with pytest.raises(RuntimeError, match='the device decoders flagged a chunk'):
            engine.collect(ticket)
        # the whole frame: decoded on the device, the same records as the block a reader had undone
        with open(path, 'r+b') as fh:
            fh.seek(offs[0])
            fh.write(frame)
        ticket = engine.submit_vbz_parts(0, [(path, cap, 0, at, [6000, 6000], framed.tobytes())], lo, hi, aut)
        rec2 = engine.collect(ticket)[0]
        assert rec2.tobytes() == rec.tobytes()
        torch.cuda.synchronize()
    finally:
        engine.close()
        os.unlink(path)

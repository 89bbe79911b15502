"""Step 3 as a multi-GPU product path: `python -m torch.distributed.run --nproc-per-node 2 -m warpstr_amd.wrapper ...` -- two
fresh ranks on the one card (gloo for the collectives, as the bench contract test does), every rank reading and calling only
its shard through the real CallerWrapper, rank 0 writing the outputs.  The files must be byte-identical to the single-rank
run's.  A one-rank RCCL group (WARPSTR_DIST_SELF_GATHER=1) runs the same collectives over the nccl backend."""
import filecmp
import json
import os
import shutil
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import GOLDEN
from warpstr_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUTPUTS = ['overview.csv', 'predictions/sequences/all.fasta', 'predictions/sequences/sequences_template.fasta',
           'predictions/sequences/sequences_reverse.fasta', 'summaries/state_similarity.csv']


def _free_port():
    from tests.helpers import free_port
    return free_port()


def _run(locus_args, ranks, extra_env=None, genotype=False):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    env.update(extra_env or {})
    tail = ['-m', 'warpstr_amd.wrapper'] + locus_args + (['--genotype'] if genotype else [])
    if ranks == 1:
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={ranks}', '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port())] + tail
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout


def _synthetic_locus(tmp_path, name, pattern, fl, n, T, seed):
    """overview.csv + flank file + an .npz of normalised segments (what --segments-npz takes)."""
    import pandas as pd
    locus = synth.make_locus(pattern, fl, seed)
    sigs, revs, _ = synth.batch(locus, n, T, seed + 1, lo=6, hi=12)
    loc = tmp_path / name
    (loc / 'expected_signals').mkdir(parents=True)
    (loc / 'expected_signals' / 'sequences.csv').write_text(
        'type,sequence\n' + ''.join(f'{k},{v}\n' for k, v in zip(
            ['left_flank_template', 'right_flank_template', 'left_flank_reverse', 'right_flank_reverse'],
            [locus.left_t, locus.right_t, locus.left_r, locus.right_r])))
    names = [f'read{i:03d}' for i in range(n + 3)]
    lens = [len(s) for s in sigs] + [100, 100, 100]
    pd.DataFrame({'read_name': names, 'run_id': 0, 'reverse': list(revs) + [False, True, False], 'saved': [1] * n + [0, 0, 0],
                  'l_start_raw': 1000, 'r_end_raw': [1000 + L - 1 for L in lens]}).to_csv(loc / 'overview.csv', index=False)
    np.savez(loc / 'signals.npz', **{nm: s for nm, s in zip(names, sigs)})
    return str(loc), locus


def _same_outputs(a, b, extra=()):
    for rel in list(OUTPUTS) + list(extra):
        assert filecmp.cmp(os.path.join(a, rel), os.path.join(b, rel), shallow=False), rel


@pytest.mark.parametrize('pattern,fl,T,extra', [('(AGC)AACAGCCGCCAC(CGC)', 20, (1400, 2200), ['predictions/complexSTR_analysis/complex_repeat_units.csv']),
                                                ('(AAAT)', 110, (2300, 3000), [])])
def test_two_ranks_write_what_one_rank_writes(tmp_path, pattern, fl, T, extra):
    """A single-slot locus (two repeat units: the complex-unit table too) and a four-slot flank-110 locus, ragged lengths:
    two ranks, each calling its cost-balanced shard, against one rank."""
    loc1, _ = _synthetic_locus(tmp_path, 'one', pattern, fl, 41, T, 5)
    loc2 = str(tmp_path / 'two')
    shutil.copytree(loc1, loc2)
    args = lambda loc: [loc, pattern, str(fl), '--segments-npz', os.path.join(loc, 'signals.npz')]
    assert '41 reads called' in _run(args(loc1), 1)
    assert '41 reads called' in _run(args(loc2), 2, {'WARPSTR_DIST_BACKEND': 'gloo'})
    _same_outputs(loc1, loc2, extra)


def test_upstream_test_case_on_two_ranks_and_on_a_one_rank_rccl_group(tmp_path):
    """The upstream test case (10 real reads of a VBZ fast5, flank-110 (AAAT) automaton): each rank opens only its own
    reads' records of the fast5 file and prepares them on the GPU from int16; one rank, two ranks (gloo) and a one-rank
    group over RCCL all write the same files, and step 4 on rank 0 reports the README's (44, 40)."""
    from warpstr_amd import fast5, overview as ov
    from warpstr_amd.wrapper import prepare_caller_only
    try:
        fast5._libs()
    except fast5.Fast5Error as e:
        pytest.skip(str(e))
    real = os.path.join(GOLDEN, 'real')
    with open(os.path.join(real, 'flanks.json')) as f:
        fj = json.load(f)
    d = tmp_path / 'test' / 'test_input' / 'test_run1' / 'fast5s'
    d.mkdir(parents=True)
    os.symlink(os.path.join(real, 'batch_0.fast5'), d / 'batch_0.fast5')
    locs = []
    for tag in ('one', 'two', 'rccl'):
        loc = prepare_caller_only(os.path.join(real, 'example.csv'), str(tmp_path / tag), base_dir=str(tmp_path))['Human_STR_1108232']
        ov.store_flanks(loc, [fj['left_template'], fj['right_template'], fj['left_reverse'], fj['right_reverse']])
        locs.append(loc)
    args = lambda loc: [loc, fj['sequence'], str(fj['flank_length'])]
    out1 = _run(args(locs[0]), 1, genotype=True)
    out2 = _run(args(locs[1]), 2, {'WARPSTR_DIST_BACKEND': 'gloo'}, genotype=True)
    out3 = _run(args(locs[2]), 1, {'WARPSTR_DIST_SELF_GATHER': '1', 'WARPSTR_DIST_BACKEND': 'nccl', 'MASTER_PORT': str(_free_port())}, genotype=True)
    for out in (out1, out2, out3):
        assert '10 reads called' in out and 'Allele lengths as given by WarpSTR: (44, 40)' in out
    _same_outputs(locs[0], locs[1])
    _same_outputs(locs[0], locs[2])
    for loc in locs[1:]:
        assert filecmp.cmp(os.path.join(locs[0], 'predictions', 'alleles.csv'), os.path.join(loc, 'predictions', 'alleles.csv'), shallow=False)

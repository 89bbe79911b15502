"""bench.py prints exactly one JSON line with the fields the driver reads (small workloads, one GPU), checks its own
results against the oracle, and its N > 1 path runs as the driver launches it (two ranks on the one GPU, gloo)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline')


def _one_line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in CONTRACT:
        assert k in d, k
    return d


def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--reads', '9000', '--samples', '900', '--steps', '3',
                          '--warmup', '1'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    d = _one_line(out)
    assert 'cpu_baseline' in d
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['unit'] == 'reads/s' and d['scaling'] == 'weak' and d['dtype'] == 'f64' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and d['config']['called_ok'] == 9000
    assert abs(d['value'] - 9000 / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and 'traffic' in r
    assert r['achieved'] > 0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert r['launches_per_step'] >= 2 and r['fill_union_ms_per_launch'] > 0
    # launch_ms is a kernel duration (mean over the launches, as rocprofv3 --stats averages it), achieved follows from it
    assert abs(r['achieved'] - r['algorithmic_bytes_per_launch'] / (r['launch_ms'] * 1e-3) / 1e9) <= 1e-6 * r['achieved']
    assert r['launch_ms'] >= r['fill_union_ms_per_launch'] * 0.999 and r['achieved_over_union'] >= r['achieved'] * 0.999
    # the union of the launch intervals does not exceed the step
    assert r['fill_union_ms_per_step'] <= d['ms_per_step'] * 1.001
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'reads/s' and c['value'] > 0 and c['cores'] >= 1 and c['sample']
    v = d['verified']
    assert v['reads'] >= 256 and v['mismatches'] == 0
    if not os.environ.get('WARPSTR_BENCH_PROFILING'):  # (set while a changed kernel's counters are being re-collected)
        assert d['valu_roofline']['counters_from'].startswith('profiles/')


@pytest.mark.parametrize('workload', ['cfg1', 'cfg5'])
def test_bench_other_workloads_keep_the_contract(workload):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', workload, '--reads', '1500', '--steps', '2',
                          '--warmup', '1', '--no-cpu-baseline'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    d = _one_line(out)
    assert d['config']['name'] == workload and d['config']['called_ok'] == 1500
    assert d['verified']['mismatches'] == 0 and d['verified']['reads'] > 0


def test_bench_from_raw_leg_reports_both_variants_and_checks_them():
    """--from-raw: int16 segments -> wsx_prepare_signals -> wsx_call_batch with both sequences -> records, from HBM and
    from pinned host memory, as a secondary field of the same one line; the leg compares its records and sequences with the
    float64 path."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--reads', '6000', '--samples', '900', '--steps', '2',
                          '--warmup', '1', '--no-cpu-baseline', '--from-raw'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    d = _one_line(out)
    f = d['from_raw']
    assert f['reads_per_s_hbm_int16'] > 0 and f['reads_per_s_host_int16'] > 0
    assert f['identical_to_f64_path']['identical'] is True and f['identical_to_f64_path']['reads'] > 0
    assert f['h2d_bytes_per_step'] == 6000 * 900 * 2


def test_default_run_carries_the_secondary_workloads():
    """The driver's command (no --reads, one GPU): besides the headline, ONE line also holds driver-timed, oracle-checked
    numbers for the flank-110 shape (cfg1), configs[4]'s share (cfg5) and the path from raw int16 segments, and the
    reference's own Python timing as a constant next to the CPU baseline."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '1'], capture_output=True,
                         text=True, timeout=900, cwd=ROOT)
    d = _one_line(out)
    if os.environ.get('WARPSTR_BENCH_PROFILING'):
        pytest.skip('profiling runs leave the secondary workloads out')
    sec = d['secondary']
    for name in ('cfg1', 'cfg5'):
        leg = sec[name]
        assert leg['ms_per_step'] > 0 and leg['value'] > 0 and leg['called_ok'] == leg['reads_per_step']
        assert leg['verified']['reads'] >= 256 and leg['verified']['mismatches'] == 0 and leg['kernels']
    assert sec['from_raw']['identical_to_f64_path']['identical'] is True and sec['from_raw']['ms_per_step_host_int16'] > 0
    ref = d['cpu_baseline']['reference_python']
    assert ref['kind'] == 'reference' and ref['value'] > 0 and ref['cores'] == 8
    assert d['config']['name'] == 'headline' and d['verified']['mismatches'] == 0
    # configs[4] at its FULL size on the one card, through the product seam: 8 loci x 2 strands x 50 000 reads in one handle
    full = sec['cfg5_full']
    assert full['reads'] == 400000 and full['called_ok'] >= 0.98 * full['reads'], full
    assert full['verified']['reads'] >= 64 and full['verified']['mismatches'] == 0
    assert full['deterministic']['files_identical'] is True
    assert full['copies_of_one_template_within_6_bases_of_their_median'] > 0.9
    assert full['workspace_bytes'] > 0 and full['samples'] > 9e8


def _free_port():
    from tests.helpers import free_port
    return free_port()


@pytest.mark.parametrize('scaling', ['weak', 'strong'])
def test_bench_two_ranks_as_the_driver_launches_it(scaling):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`: fresh child processes (this process never
    touches the GPU), both ranks on the one card, gloo instead of RCCL (one GPU cannot host two RCCL ranks)."""
    env = dict(os.environ, WARPSTR_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    reads = 6000 if scaling == 'weak' else 9001  # strong: an odd total, so the shards differ in size
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--reads', str(reads), '--samples', '900', '--scaling', scaling, '--no-cpu-baseline']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    d = _one_line(out)
    assert d['n_gpus'] == 2 and d['scaling'] == scaling
    assert 'all_gather' in d['config']['results_gather']
    if scaling == 'weak':
        assert d['config']['reads_per_gpu'] == reads and d['config']['called_ok'] == 2 * reads
        assert abs(d['value'] - 2 * reads / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    else:
        assert d['config']['reads_total'] == reads and d['config']['called_ok'] == reads
        assert d['config']['reads_per_gpu'] in (reads // 2, reads // 2 + 1)
        assert 'configs[3]' in d['config']['workload']
        assert abs(d['value'] - reads / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    assert d['verified']['mismatches'] == 0


def _bare_env(**kw):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **kw)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


@pytest.mark.parametrize('scaling', ['weak', 'strong'])
def test_bare_gpus_2_starts_its_own_ranks(scaling):
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset): bench.py starts torch.distributed.run as a child
    before anything touches the GPU and relays rank 0's one line and the exit code.  The line proves what ran: world size and
    backend as torch.distributed reports them, one entry per rank (own process, device, clock), and the records the step's
    all-gather delivered equal what every rank computed itself.  Both ranks on the one card, gloo."""
    reads = 5000 if scaling == 'weak' else 7001
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--reads', str(reads),
                          '--samples', '900', '--scaling', scaling, '--no-cpu-baseline'], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=_bare_env(WARPSTR_BENCH_BACKEND='gloo'))
    d = _one_line(out)
    assert d['n_gpus'] == 2 and d['scaling'] == scaling and d['verified']['mismatches'] == 0
    r = d['ranks']
    assert r['world_size'] == 2 and r['backend'] == 'gloo' and 'itself' in r['launched_by']
    assert [e['rank'] for e in r['per_rank']] == [0, 1] and len({e['pid'] for e in r['per_rank']}) == 2
    assert all(e['ms_per_step'] > 0 and e['reads'] > 0 for e in r['per_rank'])
    assert abs(d['ms_per_step'] - r['ms_per_step_max_over_ranks']) < 1e-9
    assert d['ms_per_step'] >= max(e['ms_per_step'] for e in r['per_rank']) * 0.999
    assert r['gathered_records_equal_every_ranks_own'] is True
    assert sum(e['reads'] for e in r['per_rank']) == (2 * reads if scaling == 'weak' else reads)


def test_bare_gpus_2_over_rccl_on_one_gpu_says_why_not():
    """RCCL wants a GPU per rank: on the one-GPU box `--gpus 2` ends non-zero with one sentence, before any rank is started."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], capture_output=True,
                         text=True, timeout=300, cwd=ROOT, env=_bare_env())
    assert out.returncode == 2 and out.stdout.strip() == ''
    assert 'needs 2 GPUs on this node, 1 visible' in out.stderr


def test_bench_four_ranks_on_the_one_card():
    """The strong-scaling line at four ranks (configs[3]'s partition at a size the card can host four times: the box allows six
    processes on its GPU, so the driver's --gpus 8 cannot be rehearsed here; the collective is gloo): every read called once,
    the all-gathered table complete, shards of 2 000-2 001 reads."""
    env = dict(os.environ, WARPSTR_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '4', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '3', '--warmup', '1',
           '--reads', '8001', '--samples', '900', '--scaling', 'strong', '--no-cpu-baseline']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    d = _one_line(out)
    assert d['n_gpus'] == 4 and d['scaling'] == 'strong' and d['config']['reads_total'] == 8001 and d['config']['called_ok'] == 8001
    assert d['config']['reads_per_gpu'] in (2000, 2001) and d['verified']['mismatches'] == 0
    assert abs(d['value'] - 8001 / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']


@pytest.mark.parametrize('scaling', ['weak', 'strong'])
def test_bench_rccl_collective_path_with_a_one_rank_group(scaling):
    """The code path N > 1 takes on real hardware -- RCCL all_gather_into_tensor of the records on a side stream, ordered
    after each step with wsx_caller_join, overlapped with the next step -- run with a one-rank RCCL group on the one GPU
    (WARPSTR_BENCH_SELF_GATHER=1)."""
    env = dict(os.environ, WARPSTR_BENCH_SELF_GATHER='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--reads', '7001', '--samples', '900', '--steps', '4',
                          '--warmup', '1', '--scaling', scaling, '--no-cpu-baseline'], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=env)
    d = _one_line(out)
    assert 'nccl' in d['config']['results_gather'] and d['scaling'] == scaling
    assert d['config']['called_ok'] == 7001 and d['verified']['mismatches'] == 0
    r = d['ranks']   # the proof fields of the N > 1 line, through RCCL itself
    assert r['world_size'] == 1 and r['backend'] == 'nccl' and r['gathered_records_equal_every_ranks_own'] is True
    assert r['per_rank'][0]['reads'] == 7001 and r['per_rank'][0]['device'] == 0

"""bench.py prints exactly one JSON line with the fields the driver reads (small workload, one GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--reads', '9000', '--samples', '900', '--steps', '3',
                          '--warmup', '1'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['unit'] == 'reads/s' and d['scaling'] == 'weak' and d['dtype'] == 'f64' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and d['config']['called_ok'] == 9000
    assert abs(d['value'] - 9000 / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and 'traffic' in r
    assert r['achieved'] > 0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert r['launches_per_step'] >= 2 and r['launch_ms'] > 0
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'reads/s' and c['value'] > 0 and c['cores'] >= 1 and c['sample']

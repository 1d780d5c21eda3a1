"""GPU: bench.py prints exactly one JSON line with the keys the driver parses (single-GPU path; the
multi-GPU launch goes through torch.distributed.run with the same script)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline'], capture_output=True, text=True, cwd=ROOT, timeout=800)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ['metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline']:
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1
    assert d['unit'] == 'images/sec' and d['higher_is_better'] is True and d['scaling'] == 'weak'
    from eo_vae_amd import ops
    assert d['vs_baseline'] is None and d['dtype'] == ops.DEFAULT_PRECISION and 'workload' in d['config']
    assert abs(d['value'] - 16 * 1000.0 / d['ms_per_step']) < 1e-2 * d['value']
    r = d['roofline']
    for k in ['bound', 'achieved', 'peak', 'unit', 'frac', 'traffic']:
        assert k in r, k
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert r['traffic'] is None or r['traffic'] > 0
    assert 'traffic_source' in r and (r['traffic'] is None or not r['traffic_source'].get('stale'))
    w = d['roofline_wgrad']      # second kernel family (3x3 weight gradients), same measurement
    assert w['bound'] == 'mfma' and 0 < w['frac'] < 1 and abs(w['frac'] - w['achieved'] / w['peak']) < 1e-3
    assert w['launches'] > 0 and 0 < w['kernel_time_share'] < 1

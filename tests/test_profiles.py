"""The committed evidence of the current round is consistent with what DESIGN.md quotes from it: `profiles/r06/kernel_table.csv` (one
kernel-trace pass per workload, tools/profile_all.sh + tools/kernel_table.py) and the driver's command's output
(`bench_default_line.json`).  No GPU: the files are data."""
import csv
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R06 = os.path.join(REPO, 'profiles', 'r06')


def _rows():
    return list(csv.DictReader(open(os.path.join(R06, 'kernel_table.csv'))))


def _row(rows, workload, prefix):
    hits = [r for r in rows if r['workload'] == workload and r['kernel'].startswith(prefix)]
    assert hits, (workload, prefix)
    return hits[0]


def test_kernel_table_recomputes_the_fractions_design_quotes():
    rows = _rows()
    # every row that carries algorithmic work: achieved = work / average launch, frac = achieved / peak
    checked = 0
    for r in rows:
        if not r['algorithmic_per_launch']:
            continue
        work, ms, peak = float(r['algorithmic_per_launch']), float(r['avg_ms']), float(r['peak'])
        scale = 1e9 if r['achieved_unit'] == 'GB/s' else 1e12
        assert float(r['achieved']) == pytest.approx(work / (ms * 1e-3) / scale, rel=3e-3), r   # (avg_ms is printed with five decimals)
        assert float(r['frac_of_peak']) == pytest.approx(float(r['achieved']) / peak, abs=2e-4), r
        checked += 1
    assert checked >= 40
    # the figures of DESIGN.md section 4 (two significant digits there; these are the rows they come from)
    expect = {('headline', 'symv_tiles_kernel<double, true'): (0.82, 0.84), ('c2', 'symv_tiles_kernel<double, true'): (0.79, 0.83),
              ('c4', 'symv_tiles_kernel<double, true'): (0.81, 0.84), ('c5', 'symv_tiles_kernel<float, true'): (0.78, 0.81),
              ('dense_n100000', 'symv_tiles_kernel<double, false'): (0.81, 0.84), ('dense_n100000', 'gemv_rows_kernel'): (0.85, 0.88),
              ('headline', 'gram_mfma_kernel<double, 2>'): (0.64, 0.68), ('stream_n100000', 'gram_stream_sym_kernel'): (0.70, 0.74),
              ('stream_n400000', 'gram_stream_sym_kernel'): (0.72, 0.75), ('chol_n50048', 'blocked Cholesky'): (0.83, 0.86),
              ('ip_c3', 'blocked Cholesky'): (0.84, 0.87)}
    for (wl, pref), (lo, hi) in expect.items():
        assert lo <= float(_row(rows, wl, pref)['frac_of_peak']) <= hi, (wl, pref)
    # the packed dense layout moves half the bytes of the row blocks for the same operator
    packed, full = _row(rows, 'dense_n100000', 'symv_tiles_kernel<double, false'), _row(rows, 'dense_n100000', 'gemv_rows_kernel')
    assert float(packed['algorithmic_per_launch']) < 0.51 * float(full['algorithmic_per_launch'])
    assert float(packed['avg_ms']) < 0.55 * float(full['avg_ms'])
    # an iteration of PG and of AdaGrad on the augmented Lagrangian at config 2's size: tile kernel + ONE closing kernel + one
    # elementwise update kernel, launched once per iteration each
    for wl, closing, update in (('c2', 'symv_reduce_kernel<8, 1>', 'pgfw_update_kernel'), ('adagrad_n20000', 'symv_reduce_kernel<8, 2>', 'al_update_kernel')):
        a, b = int(_row(rows, wl, closing)['launches']), int(_row(rows, wl, update)['launches'])
        assert a == b and float(_row(rows, wl, closing)['avg_ms']) < 0.020 and float(_row(rows, wl, update)['avg_ms']) < 0.008


def test_pmc_traffic_matches_the_algorithmic_bytes():
    rec = json.load(open(os.path.join(R06, 'pmc_traffic_n100k_pg.json')))
    k = next(v for name, v in rec['kernels'].items() if name.startswith('symv_tiles'))
    alg = float(_row(_rows(), 'headline', 'symv_tiles_kernel<double, true')['algorithmic_per_launch'])
    assert 0.99 < k['hbm_bytes'] / alg < 1.02     # no wasted re-reads


def test_the_committed_driver_line_parses_and_says_which_placement_it_is():
    lines = [l for l in open(os.path.join(R06, 'bench_default_line.json')).read().splitlines() if l.strip()]
    assert len(lines) == 2 and all(len(l) < 4096 for l in lines)
    first, last = json.loads(lines[0]), json.loads(lines[1])
    assert 'headline record only' in first['state'] and 'state' not in last and first['value'] == last['value']
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in first and k in last
    assert last['config']['workload'] == 'svc_hinge_rbf_pg_dual_n100000_d128' and last['dtype'] == 'f64'
    assert last['config']['panel_placement_ms'] and last['roofline']['frac_first_placement'] > 0.7
    assert 0.80 < last['roofline']['frac'] < 0.85 and last['roofline']['bound'] == 'hbm' and last['roofline']['traffic']
    assert last['cpu_baseline']['kind'] == 'port' and last['cpu_baseline']['cores'] >= 1
    kk = json.load(open(os.path.join(R06, 'time_to_kkt_headline.json')))
    assert kk['ip_headline']['status'] == 'optimal' and kk['as_headline']['status'] == 'optimal'
    assert last['side']['ip_headline_s'] == pytest.approx(kk['ip_headline']['value'], rel=1e-4)
    assert last['side']['as_headline_s'] == pytest.approx(kk['as_headline']['value'], rel=1e-4)
    assert abs(kk['ip_headline']['f'] - kk['as_headline']['f']) < 1e-7 * abs(kk['ip_headline']['f'])   # the two solvers agree on the optimum

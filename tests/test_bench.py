"""bench.py's launch logic: `--gpus N` from a plain shell starts N fresh rank processes itself (the parent never touches the
GPU); a missing RCCL communicator is an error, not a silent host fallback."""
import json
import os
import subprocess
import sys

import pytest

from conftest import set_hooks, hooks_env, hook_value

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, 'bench.py')


def _run(args, timeout=600, env=None, line='full'):
    """`--line full`: the complete record on stdout (what these tests read); `line='compact'` is the driver's view."""
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args + ['--line', line], capture_output=True, text=True, timeout=timeout, env=e, cwd=REPO)


def test_presets_and_argument_checks():
    sys.path.insert(0, REPO)
    import bench
    a = bench.parse([])
    assert (a.n, a.d, a.solver, a.task, a.kernel, a.storage, a.gpus) == (100000, 128, 'pg', 'svc', 'rbf', 'f64', 1)
    a = bench.parse(['--config', 'c5', '--gpus', '8'])
    assert (a.n, a.d, a.solver, a.storage, a.gpus) == (250000, 256, 'ascg', 'f32', 8)
    a = bench.parse(['--config', 'c4', '--samples', '20000'])
    assert (a.n, a.task, a.kernel, a.solver) == (20000, 'svr', 'poly', 'fw')
    r = _run(['--gpus', '2', '--solver', 'ip'], timeout=120)
    assert r.returncode != 0 and 'replicas only' in r.stderr


def test_self_launch_without_a_gpu_fails_fast_and_loudly():
    """No GPU in the build container: both spawned ranks fail to create a device context; the parent reports it and returns
    non-zero within seconds — no hang, no fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('needs a box without a GPU')
    r = _run(['--gpus', '2', '--samples', '2000', '--features', '8', '--steps', '2', '--warmup', '1', '--no-cpu'], timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ''


@pytest.mark.gpu
def test_self_launch_two_ranks_host_exchange_on_one_gpu():
    """`python bench.py --gpus 2` from a plain shell: two fresh rank processes (both on GPU 0 of this one-GPU box, host
    exchange), ONE JSON line from rank 0, same iterates as the single-process run (bit-identical objective)."""
    common = ['--samples', '6000', '--features', '16', '--steps', '8', '--warmup', '2', '--no-cpu', '--kkt', 'none']
    one = _run(common + ['--gpus', '1'])
    assert one.returncode == 0, one.stderr
    two = _run(common + ['--gpus', '2', '--exchange', 'host'])
    assert two.returncode == 0, two.stderr
    lines = [l for l in two.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    a, b = json.loads(one.stdout.strip().splitlines()[-1]), json.loads(lines[0])
    assert b['n_gpus'] == 2 and b['config']['exchange'] == 'host' and b['config']['sym_exchange'] == 'gather'
    assert b['config']['rccl_ranks'] == 0 and b['steps_done'] == 8
    assert a['f_last'] == b['f_last'] and a['kkt_resid_last'] == b['kkt_resid_last']
    for rec in (a, b):
        assert rec['roofline']['bound'] == 'hbm' and 0 < rec['roofline']['frac'] < 1.2
        assert rec['roofline']['frac_survey_8d_bytes'] > rec['roofline']['frac']


@pytest.mark.gpu
def test_self_launch_without_torch_rendezvous():
    """BQ_RENDEZVOUS=socket: the ranks meet over the package's own TCP communicator (what a host without torch gets); the N > 1
    line carries every rank's share and timings."""
    common = ['--samples', '6000', '--features', '16', '--steps', '6', '--warmup', '2', '--no-cpu', '--kkt', 'none']
    r = _run(common + ['--gpus', '2', '--exchange', 'host'], env={'BQ_RENDEZVOUS': 'socket'})
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec['n_gpus'] == 2 and rec['steps_done'] == 6 and rec['config']['exchange'] == 'host'
    pr = rec['per_rank']
    assert [p['rank'] for p in pr] == [0, 1] and sum(p['rows'] for p in pr) == 6000
    nb = -(-6000 // 256)
    assert sum(p['tiles'] for p in pr) == nb * (nb + 1) // 2
    assert all(p['symv_tiles_ms'] > 0 and p['ms_per_step'] > 0 for p in pr)
    ec = rec['exchange_compare']   # both closing collectives timed in the one run (the other one outside the timed region)
    assert set(ec) == {'gather', 'allreduce'} and ec['allreduce']['steps'] == 10
    assert all(v['exchange_ms_per_product'] > 0 and v['ms_per_step'] > 0 for v in ec.values())


@pytest.mark.gpu
def test_rccl_that_cannot_be_created_is_an_error_not_a_fallback():
    """Two ranks on ONE device cannot form an RCCL communicator: exit code 3, no JSON line; with --allow-host-exchange the
    run falls back and says so."""
    common = ['--samples', '4000', '--features', '8', '--steps', '3', '--warmup', '1', '--no-cpu', '--kkt', 'none', '--gpus', '2']
    env = {'NCCL_DEBUG': 'WARN'}     # RCCL's own chatter goes to fd 1: it must not reach the JSON channel either
    r = _run(common, env=env)
    assert r.returncode == 3 and r.stdout.strip() == '', (r.returncode, r.stdout, r.stderr[-2000:])
    assert 'RCCL context unavailable' in r.stderr
    r = _run(common + ['--allow-host-exchange'], env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec['config']['exchange'] == 'host' and rec['config']['rccl_ranks'] == 0


@pytest.mark.gpu
def test_launched_by_torch_distributed_run_like_the_driver(tmp_path):
    """The driver's N > 1 command: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...  (two ranks on GPU 0 of this box, host exchange): ONE JSON line on stdout."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), BENCH, '--gpus', '2', '--exchange', 'host', '--samples', '5000', '--features', '16',
           '--steps', '6', '--warmup', '2', '--no-cpu', '--kkt', 'none', '--records-file', str(tmp_path / 'full.json')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, r.stdout      # the driver's view: ONE compact line (it keeps a 12.8 KB tail)
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps_done'] == 6 and rec['config']['exchange'] == 'host'
    assert rec['roofline']['bound'] == 'hbm' and rec['roofline']['frac'] > 0 and rec['cpu_baseline'] is None
    assert len(rec['per_rank']['symv_tiles_ms']) == 2 and set(rec['exchange_compare']) == {'gather', 'allreduce'}
    full = json.load(open(tmp_path / 'full.json'))                   # ... and the uncut record beside it
    assert full['value'] == pytest.approx(rec['value'], rel=1e-5) and [p['rank'] for p in full['per_rank']] == [0, 1]


def test_default_line_is_composed_from_child_records(monkeypatch, capsys, tmp_path):
    """`python bench.py` with no workload flag: the parent composes ONE line from the headline child's record and the side
    records (configs / shares / collective floor), never touching HIP itself; a failing or over-budget side record is reported
    inside the line, not fatal."""
    sys.path.insert(0, REPO)
    import bench
    args = bench.parse(['--gpus', '1', '--steps', '20', '--warmup', '5'])
    assert args.default_workload and args.records == 'all' and args.cpu_stream_iters == 1
    assert bench.parse(['--samples', '5000']).records == 'none'
    assert args.line == 'compact'
    args.records_file = str(tmp_path / 'bench_records.json')
    calls = []

    def share_rec(workload, gs, ms):
        parts = []
        for G in gs:
            shares = [{'rank': k, 'tiles': 100, 'strips': 20, 'symv_tiles_ms': ms / G, 'symv_frac_of_8TBs': 0.8, 'ms_per_step': ms / G + 0.05,
                       'fixed_cost_ms': 0.05} for k in range(G)]
            parts.append({'G': G, 'shares': shares, 'slowest_share_ms_per_step': ms / G + 0.05, 'min_symv_frac_of_8TBs': 0.8,
                          'predicted_iter_per_s_no_exchange': 1e3 / (ms / G + 0.05), 'predicted_iter_per_s': 1e3 / (ms / G + 0.1)})
        return {'what': 'share_timing', 'config': {'workload': workload, 'steps': 30}, 'assumed_exchange_us': 50.0, 'partitions': parts}

    def fake_child(argv, timeout):
        calls.append((list(argv), timeout))
        if '--dense' in argv:
            lay = lambda v, f: {'value': v, 'roofline': {'frac': f, 'avg_launch_ms': 0.3}}
            return {'what': 'dense_quadratic', 'n': 20000, 'layouts': {'packed': lay(3500.0, 0.81), 'rows': lay(2000.0, 0.85)}}, None, 5.0
        if '--collective-floor' in argv:
            return {'what': 'collective_floor', 'headline': {'gather_8_segments': {'mean_us': 5.0}, 'allreduce': {'mean_us': 4.0}}}, None, 1.0
        if '--emulate-shares' in argv:
            if 'c5' in argv:
                return None, 'exit code 1, no JSON line', 2.0
            return share_rec('c4' if 'c4' in argv else 'headline', [int(g) for g in argv[argv.index('--emulate-shares') + 1].split(',')], 6.0), None, 3.0
        if '--solver' in argv and argv[argv.index('--solver') + 1] == 'as':
            return {'metric': 'time_to_kkt_tol', 'value': 10.0, 'unit': 's', 'iterations': 22897, 'status': 'optimal', 'f': -57.5,
                    'config': {'workload': 'svc_hinge_rbf_as_dual_n20000_d64'}, 'cpu_baseline': None}, None, 12.0
        if '--config' in argv:
            cfg = argv[argv.index('--config') + 1]
            rec = {'metric': 'dual_qp_iterations_per_sec', 'value': 10.0, 'ms_per_step': 100.0, 'config': {'workload': cfg},
                   'roofline': {'frac': 0.8, 'avg_launch_ms': 9.0}, 'cpu_baseline': None}
            if cfg == 'c5':
                rec['inner_products_per_step'] = 10.0
            return rec, None, 4.0
        return {'metric': 'dual_qp_iterations_per_sec', 'value': 160.0, 'ms_per_step': 6.25, 'n_gpus': 1, 'steps': 20, 'warmup': 5,
                'roofline': {'frac': 0.82, 'avg_launch_ms': 6.1}, 'cpu_baseline': {'value': 0.2}}, None, 100.0

    monkeypatch.setattr(bench, '_run_child', fake_child)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '1', '--steps', '20', '--warmup', '5'])
    bench.orchestrate(args)
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    # the LAST stdout line is the compact record (r04's 25.6 KB line overflowed the driver's 12.8 KB tail); the one before it is the
    # same record as it stood when the headline child returned (a run cut during the side records leaves THAT as its last line)
    assert len(lines) == 2 and all(len(l) < bench.LINE_LIMIT == 4096 for l in lines)
    first = json.loads(lines[0])
    assert first['value'] == 160.0 and first['roofline']['frac'] == 0.82 and first['cpu_baseline']['value'] == 0.2
    assert 'side' not in first and 'headline record only' in first['state']
    line = json.loads(lines[-1])
    assert 'state' not in line
    assert line['side']['dense_c2_packed_rows_iter_s'] == [3500.0, 2000.0] and line['side']['dense_c2_packed_rows_frac'] == [0.81, 0.85]
    assert line['value'] == 160.0 and line['steps'] == 20 and line['warmup'] == 5 and line['n_gpus'] == 1
    assert line['roofline']['frac'] == 0.82 and line['cpu_baseline']['value'] == 0.2
    assert line['side']['c2_iter_s'] == 10.0 and line['side']['c5_products_per_outer_it'] == 10.0 and line['side']['as_c2_s'] == 10.0
    assert line['side']['predicted_x_at_2_4_8'][2] == pytest.approx(1e3 / (6.0 / 8 + 0.1) / 160.0, rel=1e-5)
    assert line['side']['collective_floor_us'] == {'gather_8_segments': 5.0, 'allreduce': 4.0}
    rec = json.load(open(args.records_file))                                      # the complete record, beside the line
    assert calls[0][0] == ['--gpus', '1', '--steps', '20', '--warmup', '5']       # the headline child runs exactly the caller's command
    assert [c[0][:2] for c in calls[1:4]] == [['--config', 'c2'], ['--config', 'c4'], ['--config', 'c5']]
    assert rec['value'] == 160.0 and rec['steps'] == 20 and rec['cpu_baseline'] == {'value': 0.2}   # headline fields untouched
    assert set(rec['configs']) == {'c2', 'c4', 'c5'} and 'cpu_baseline' not in rec['configs']['c2']
    head = rec['shares']['headline']['partitions']
    assert [p['G'] for p in head] == [1, 2, 4, 8] and head[0]['predicted_iter_per_s'] == 160.0
    assert head[3]['predicted_speedup_vs_1'] == pytest.approx(head[3]['predicted_iter_per_s'] / 160.0)
    assert head[3]['predicted_iter_per_s_at_collective_floor'] == pytest.approx(1e3 / (6.0 / 8 + 0.05 + 0.005))
    assert rec['shares']['c4_over_4']['partitions'][0]['one_gpu_ms_per_step'] == 100.0
    assert 'error' in rec['shares']['c5_over_8']                                  # a failing side record stays inside the line
    assert rec['collective_floor_us']['headline']['allreduce']['mean_us'] == 4.0
    assert rec['records']['requested'] == list(bench.SIDE_RECORDS)
    assert rec['time_to_kkt']['as_config2_shape']['iterations'] == 22897 and rec['time_to_kkt']['as_config2_shape']['status'] == 'optimal'
    # a budget that is already spent: the headline still runs, every side record is skipped with the reason
    calls.clear()
    args.budget_s = 1.0
    bench.orchestrate(args)
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    rec = json.load(open(args.records_file))
    assert len(calls) == 1 and all('skipped' in v for v in rec['configs'].values())
    assert line['value'] == 160.0 and 'skipped' not in line and all('budget left' in line['side'][f'{c}_iter_s'] for c in ('c2', 'c4', 'c5'))


def test_a_run_cut_after_the_headline_child_leaves_a_parseable_last_line(monkeypatch, capsys, tmp_path):
    """The driver's limit (or a slow box) can end the default run while the side records are being measured — round 4 lost its whole
    record that way.  The line is therefore printed as soon as the headline record exists: here the first side record's child never
    returns (the parent is interrupted inside it), and the last stdout line still parses with every contract field."""
    sys.path.insert(0, REPO)
    import bench
    args = bench.parse(['--gpus', '1', '--steps', '20', '--warmup', '5'])
    args.records_file = str(tmp_path / 'bench_records.json')
    head = {'metric': 'dual_qp_iterations_per_sec', 'value': 163.0, 'unit': 'iter/s', 'n_gpus': 1, 'steps': 20, 'warmup': 5, 'ms_per_step': 6.13,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'svc_hinge_rbf_pg_dual_n100000_d128', 'n': 100000, 'd': 128, 'panel_placement_ms': [6.3312345, 6.0912345],
                       'placement_budget': 'steady state: up to 5 s (SVC.fit: 2 % of max_iter products, 0.2 s at least)', 'device': 'X (gfx950)'},
            'roofline': {'bound': 'hbm', 'achieved': 6650.0, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.83, 'traffic': None, 'avg_launch_ms': 6.06,
                         'frac_first_placement': 0.795},
            'cpu_baseline': {'value': 0.23, 'unit': 'iter/s', 'cores': 64, 'kind': 'port', 'sample': 'oracle PG'}}
    calls = []

    def child(argv, timeout):
        calls.append(list(argv))
        if len(calls) == 1:
            return dict(head), None, 100.0
        raise KeyboardInterrupt   # the cut: what a SIGINT / the end of the caller's patience looks like from inside the parent

    monkeypatch.setattr(bench, '_run_child', child)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '1', '--steps', '20', '--warmup', '5'])
    with pytest.raises(KeyboardInterrupt):
        bench.orchestrate(args)
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(calls) == 2 and len(lines) == 1
    line = json.loads(lines[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line
    assert line['value'] == 163.0 and line['roofline']['frac'] == 0.83 and line['cpu_baseline']['value'] == 0.23
    # the line says which placement it is and what the first allocation would have read
    assert line['config']['panel_placement_ms'] == [6.331, 6.091] and line['config']['placement_budget'] == 'steady state'
    assert line['roofline']['frac_first_placement'] == 0.795
    assert 'headline record only' in line['state']
    assert json.load(open(args.records_file))['value'] == 163.0


def test_compact_line_stays_under_the_limit_whatever_the_record_holds():
    """The committed r04 record (25.6 KB: the one the driver could not keep) compacts to < 4 KB with every contract field; a record
    bloated far beyond it still does (summaries are dropped and named, contract fields never)."""
    sys.path.insert(0, REPO)
    import bench
    full = json.load(open(os.path.join(REPO, 'profiles', 'r04', 'bench_default_line_final.json')))
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full, os.path.join(REPO, 'bench_records.json'))
    assert len(json.dumps(line)) < 4096 and 'dropped' not in line
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line
    assert line['config']['workload'] == 'svc_hinge_rbf_pg_dual_n100000_d128'
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'algorithmic_bytes_per_launch', 'bytes_basis'} <= set(line['roofline'])
    assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(line['cpu_baseline'])
    assert line['side']['c5_outer_it_s'] == pytest.approx(4.68469) and line['side']['ip_c3_s'] == pytest.approx(37.4363)
    full['per_rank'] = [{'rank': k, 'tiles': 10 ** 9 + k, 'ms_per_step': 1.2345678 + k, 'symv_tiles_ms': 0.7654321 + k,
                         'exchange_ms_per_product': 0.0123 + k} for k in range(64)]
    full['fixed_cap'].update({f'x{k}': {'iterations': k, 'proj_grad_norm_2': 1.5, 'wall_s': 2.5} for k in range(200)})
    line = bench.compact_line(full, None)
    assert len(json.dumps(line)) < 4096 and line['dropped'] and line['roofline']['frac'] and line['cpu_baseline']['value']


def test_default_line_without_a_gpu_fails_loudly():
    """No GPU here: the headline child cannot create a context; the parent prints no JSON line and returns non-zero."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('needs a box without a GPU')
    r = _run(['--steps', '2', '--warmup', '1', '--no-cpu', '--kkt', 'none'], timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ''
    assert 'headline run failed' in r.stderr


@pytest.mark.gpu
def test_fixed_cap_and_collective_floor_records():
    """The two record kinds the default line gathers from children of its own: `--fixed-cap` (iterations, f and the
    projected-gradient norm computed the same way for every solver) and `--collective-floor` (one-rank RCCL communicator)."""
    r = _run(['--config', 'c2', '--fixed-cap', '30'])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    run = rec['runs']['c2']
    assert rec['what'] == 'fixed_cap' and rec['cap'] == 30 and run['iterations'] == 30 and run['status'] == 'stopped'
    assert run['proj_grad_norm_2'] > 0 and run['f'] == pytest.approx(run['f_solver_last_record'], rel=1e-9)
    assert run['n_at_lower'] + run['n_at_upper'] + run['n_sv_alpha_gt_1e-6'] >= run['dual_dim'] - run['n_at_upper']
    r = _run(['--collective-floor'], timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec['rccl_ranks'] == 1
    for key, n in (('headline', 100000), ('c5', 250000)):
        ln = -(-n // 256) * 256
        assert rec[key]['gather_8_segments']['bytes'] == 8 * ln * 8 and rec[key]['allreduce']['bytes'] == ln * 8
        assert 0 < rec[key]['gather_8_segments']['min_us'] <= rec[key]['gather_8_segments']['mean_us'] < 1e4


@pytest.mark.gpu
def test_products_that_returned_on_the_done_flag_are_not_counted():
    """ActiveSetCG enqueues one inner iteration more than it needs per solve (the `done` flag is looked at one iteration late);
    that launch returns at once and says so itself (bq_prof_skip_arg).  The profiled launch count of the bench record is exactly
    the number of products the solver needed — not a guess from durations."""
    if hook_value('as_cg_incq') == '0' or hook_value('as_cg_colq') == '0':
        pytest.skip('with the product-free bookkeeping switched off every outer iteration has products of its own')
    r = _run(['--config', 'c5', '--samples', '6000', '--features', '32', '--steps', '6', '--warmup', '2', '--no-cpu', '--kkt', 'none'])
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    inner = rec['inner_products_per_step'] * rec['steps_done']
    assert rec['steps_done'] == 6 and inner > 6
    # (a start product that is really needed — more than 16 variables bound at once — is a product too: none or one here)
    assert round(inner) <= rec['roofline']['launches'] <= round(inner) + 2, (rec['roofline']['launches'], inner)


def test_c5_projection_reads_the_committed_scaling():
    """time_to_kkt.c5_projected: iterations / n of config 5's workload run to 'optimal' at the largest committed size
    (profiles/rNN/c5_outer_iterations_scaling.json, tools/c5_scaling.py) x n x the measured time per outer iteration — labelled a
    projection; the compact line carries it as side.c5_projected_s."""
    sys.path.insert(0, REPO)
    import bench
    rec = bench.c5_projection(250000, 256, 112.0)
    assert rec['kind'].startswith('projected') and rec['from_n'] >= 40000 and 0.9 < rec['iterations_per_n'] < 1.0
    assert rec['outer_iterations_projected'] == pytest.approx(rec['iterations_per_n'] * 250000)
    assert rec['value'] == pytest.approx(rec['outer_iterations_projected'] * 0.112)
    assert bench.c5_projection(250000, 17, 112.0) is None          # no scaling run for another d
    full = json.load(open(os.path.join(REPO, 'profiles', 'r05', 'bench_default_records.json')))
    line = bench.compact_line(full, None)
    assert line['side']['c5_projected_s'] == pytest.approx(full['time_to_kkt']['c5_projected']['value'], rel=1e-5)
    assert line['side']['c5_products_per_outer_it'] <= 5.5 and line['side']['c5_outer_it_s'] > 8.0
    assert len(json.dumps(line)) < 4096

#!/usr/bin/env python3
"""profiles/rNN/kernel_table.csv: one row per (workload, kernel) of the per-kernel summaries tools/profile_all.sh leaves
(<workload>_kernel_stats.csv, from rocprofv3 --kernel-trace), with the ALGORITHMIC bytes or flops of one launch where the kernel
has a closed form for them, the rate that gives and its fraction of the bound's peak (HBM 8 TB/s; fp64 MFMA 78.6 TFLOP/s) — the
numbers DESIGN.md section 4 quotes can be recomputed from this file.

    python tools/kernel_table.py DIR > DIR/kernel_table.csv
"""
import csv
import glob
import json
import os
import re
import sys

HBM, MFMA = 8000.0, 78.6   # GB/s, TFLOP/s (MI355X_MICROARCH.md)
T = 256

WORKLOADS = {   # name -> n, d, storage bytes, dual dim factor
    'headline': (100000, 128, 8), 'c2': (20000, 64, 8), 'c4': (100000, 128, 8), 'c5': (250000, 256, 4),
    'dense_n100000': (100000, 0, 8), 'dense_n20000': (20000, 0, 8), 'stream_n100000': (100000, 128, 8),
    'stream_n400000': (400000, 128, 8), 'adagrad_n20000': (20000, 64, 8), 'adagrad_n100000': (100000, 128, 8),
    'ip_c3': (50000, 128, 8), 'as_c2': (20000, 64, 8), 'chol_n50048': (50048, 0, 8), 'smo_n100000': (100000, 128, 8),
}


def tiles(n):
    nb = -(-n // T)
    return nb, nb * (nb + 1) // 2


def gram_lower_tiles(n):   # 128 x 128 tiles of a packed build: tile row r (128 rows) holds the columns of its 256-tile row
    n128 = -(-n // 128)
    return sum(min(n128, 2 * (r // 2 + 1)) for r in range(n128))


def work(workload, kernel):
    """(bound, algorithmic work per launch, unit) or None"""
    if workload not in WORKLOADS:
        return None
    n, d, esz = WORKLOADS[workload]
    nb, nt = tiles(n)
    if kernel.startswith('symv_tiles_kernel'):
        s = 4 if '<float' in kernel else 8
        return 'hbm', nt * (T * T * s + T * 8) + (nt // 8 + nb) * T * 8 + 2 * n * 8, 'B'
    if kernel.startswith('gemv_rows_kernel'):
        ld = -(-n // 1024) * 1024
        s = 4 if 'f4' in kernel or 'float' in kernel else 8
        return 'hbm', n * ld * s + 3 * n * 8, 'B'
    if kernel.startswith('symv_reduce_kernel'):   # the slab entries of every output block once + the epilogue's vectors
        return 'hbm', (nt + nt // 8 + nb) * T * 8 + 10 * n * 8, 'B'
    if kernel.startswith('gram_mfma_kernel') and d:
        dp = -(-d // 16) * 16
        return 'mfma', 2.0 * gram_lower_tiles(n) * 128 * 128 * dp, 'flop'
    if kernel.startswith('gram_stream_sym_kernel') and d:
        dp = -(-d // 16) * 16
        t128 = -(-n // 128)
        return 'mfma', 2.0 * (t128 * (t128 + 1) // 2) * 128 * 128 * dp, 'flop'
    if kernel.startswith(('pgfw_update_kernel', 'al_update_kernel')):
        return 'hbm', (7 if kernel.startswith('pgfw') else 12) * n * 8, 'B'
    return None


def main(d):
    w = csv.writer(sys.stdout)
    # launches / avg_ms: the launches that did their work (rocpd_stats.py: CallsFull / AverageFullNs), not those that returned on a `done` flag
    w.writerow(['workload', 'kernel', 'launches', 'avg_ms', 'min_ms', 'share_of_gpu_time_pct', 'bound', 'algorithmic_per_launch', 'unit',
                'achieved', 'achieved_unit', 'peak', 'frac_of_peak', 'bench_value', 'bench_unit'])
    for path in sorted(glob.glob(os.path.join(d, '*_kernel_stats.csv'))):
        workload = os.path.basename(path)[:-len('_kernel_stats.csv')]
        bench = {}
        try:
            lines = [l for l in open(os.path.join(d, workload + '.json')).read().splitlines() if l.strip().startswith('{')]
            bench = json.loads(lines[-1]) if lines else {}
        except (OSError, ValueError):
            pass
        rows = list(csv.DictReader(open(path)))
        chol_ns = sum(float(r['TotalDurationNs']) for r in rows if re.match(r'(void )?(syrk_|trsm_gemm|potrf_diag)', r['Name']))
        for r in rows:
            name = r['Name'].replace('void ', '')
            short = re.sub(r'\(.*', '', name)[:70]
            if float(r['Percentage']) < 0.05 and not short.startswith(('symv', 'gemv', 'gram', 'pgfw', 'al_')):
                continue
            # kernels enqueued behind a solver's `done` flag: the launches that did their work (rocpd_stats.py: CallsFull / AverageFullNs)
            flagged = short.startswith(('symv_', 'gemv_rows', 'gram_stream', 'stream_sym', 'as_', 'pc2_', 'al_', 'pgfw_', 'finish_'))
            if flagged and r.get('CallsFull'):
                calls, avg = int(float(r['CallsFull'])), float(r['AverageFullNs']) / 1e6
            else:
                calls, avg = int(float(r['Calls'])), float(r['AverageNs']) / 1e6
            mn = float(r['MinNs']) / 1e6
            k = work(workload, short)
            if k:
                bound, alg, unit = k
                if bound == 'hbm':
                    ach, au, peak = alg / (avg * 1e-3) / 1e9, 'GB/s', HBM
                else:
                    ach, au, peak = alg / (avg * 1e-3) / 1e12, 'TFLOP/s', MFMA
                w.writerow([workload, short, calls, f'{avg:.5f}', f'{mn:.5f}', r['Percentage'], bound, f'{alg:.6g}', unit, f'{ach:.5g}', au, peak,
                            f'{ach / peak:.4f}', bench.get('value'), bench.get('unit')])
            else:
                w.writerow([workload, short, calls, f'{avg:.5f}', f'{mn:.5f}', r['Percentage'], '', '', '', '', '', '', '', bench.get('value'), bench.get('unit')])
        # the factorisation's kernels OVERLAP (the diagonal blocks and the panel solves of the next block column run on a second
        # stream beside the trailing update), so their durations do not add up to its time: the row quotes the HIP-event time of
        # the whole factorisation from the bench record of the same traced run
        roof = bench.get('roofline') or {}
        ms = bench.get('factor_ms') or roof.get('avg_factor_ms')
        if ms and workload in WORKLOADS and workload not in ('chol_n50048', 'ip_c3'):   # the order varies with the free set: no flop count
            w.writerow([workload, 'base factorisations of the kept-factor ActiveSet (HIP events around each; the order varies)', roof.get('factorisations', ''),
                        f'{ms:.4f}', '', f'{100.0 * chol_ns / sum(float(r["TotalDurationNs"]) for r in rows):.1f}', 'mfma', '', '', '', '', '', '', bench.get('value'), bench.get('unit')])
        elif ms and workload in WORKLOADS:
            n = WORKLOADS[workload][0]
            tf = n ** 3 / 3.0 / (ms * 1e-3) / 1e12
            w.writerow([workload, 'blocked Cholesky (syrk_* + trsm_gemm + potrf_diag128; HIP events around one factorisation)',
                        roof.get('factorisations', 1), f'{ms:.4f}', '', f'{100.0 * chol_ns / sum(float(r["TotalDurationNs"]) for r in rows):.1f}', 'mfma',
                        f'{n ** 3 / 3.0:.6g}', 'flop', f'{tf:.5g}', 'TFLOP/s', MFMA, f'{tf / MFMA:.4f}', bench.get('value'), bench.get('unit')])


if __name__ == '__main__':
    main(sys.argv[1])

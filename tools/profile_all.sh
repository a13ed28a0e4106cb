#!/bin/bash
# One profile pass of every kernel DESIGN.md section 4 quotes, with the library as built.  On the GPU box:
#   tools/profile_all.sh OUTDIR STAGE...        stages: product dense chol ip_c3 stream al as_c2 smo gram_counters
# Every stage = one rocprofv3 --kernel-trace --stats run of the command that DESIGN / bench.py quote (python3 itself after `--`), its
# per-kernel summary as gpurun_out/OUTDIR/<workload>_kernel_stats.csv and the bench record beside it; the headline also gets the
# FETCH_SIZE / WRITE_SIZE passes (tools/profile_bench.sh).  tools/kernel_table.py turns the summaries into kernel_table.csv.
# Long fits without a tracer (time-to-KKT at the headline size): tools/profile_kkt_headline.sh.
set -o pipefail
dir=$1; shift
out=gpurun_out/$dir
mkdir -p "$out"
export TMPDIR=/tmp
trace() {   # trace NAME command...   (a failing stage is reported and does not stop the others: each is its own evidence)
    name=$1; shift
    start=$(date +%s)
    rocprofv3 --kernel-trace --stats -d "$out/trace_$name" -- "$@" > "$out/$name.json" 2> "$out/$name.err"
    rc=$?
    db=$(find "$out/trace_$name" -name '*_results.db' | head -1)
    if [ $rc -ne 0 ] || [ -z "$db" ]; then echo "[profile_all] $name FAILED rc=$rc"; tail -n 5 "$out/$name.err"; rm -rf "$out/trace_$name"; return 0; fi
    python3 tools/rocpd_stats.py "$db" > "$out/${name}_kernel_stats.csv"
    rm -rf "$out/trace_$name"
    echo "[profile_all] $name ok $(( $(date +%s) - start ))s: $(head -2 "$out/${name}_kernel_stats.csv" | tail -1 | cut -c1-150)"
}
common=(--no-cpu --kkt none --records none --line full)
for stage in "$@"; do
  case $stage in
    product)   # the tile kernel and the closing kernels: headline (+ PMC traffic), config 2, config 4, config 5
      tools/profile_bench.sh "$dir/headline" --gpus 1 --steps 20 --warmup 5 > "$out/headline_profile_bench.txt" 2>&1 || tail -5 "$out/headline_profile_bench.txt"
      cp "$out/headline/kernel_stats.csv" "$out/headline_kernel_stats.csv" 2>/dev/null
      cp "$out/headline/bench_under_rocprof.json" "$out/headline.json" 2>/dev/null
      cp "$out/headline/pmc_traffic.json" "$out/pmc_traffic_n100k_pg.json" 2>/dev/null
      echo "[profile_all] headline: $(head -2 "$out/headline_kernel_stats.csv" | tail -1 | cut -c1-150)"
      trace c2 python3 bench.py --config c2 --steps 400 --warmup 20 "${common[@]}"
      trace c4 python3 bench.py --config c4 --steps 30 --warmup 3 "${common[@]}"
      trace c5 python3 bench.py --config c5 --steps 10 --warmup 2 "${common[@]}" ;;
    dense)     # dense Quadratic: packed lower triangle (symv_tiles_kernel<double, false>) against row blocks (gemv_rows_kernel)
      trace dense_n100000 python3 bench.py --dense --samples 100000 --steps 20 --warmup 3
      trace dense_n20000 python3 bench.py --dense --samples 20000 --steps 300 --warmup 10 ;;
    chol)      # the blocked Cholesky alone, order 50 048 (config 3's H)
      trace chol_n50048 python3 tools/bench_extra.py chol --n 50048 ;;
    ip_c3)     # InteriorPoint to its stop test at config 3 (Gram build, Newton systems, factorisations, sweeps)
      trace ip_c3 python3 bench.py --solver ip --samples 50000 --features 128 --no-cpu ;;
    stream)    # the streamed product (Gram tiles recomputed on the matrix cores) at n = 100 000 and n = 400 000
      trace stream_n100000 python3 bench.py --storage stream --steps 6 --warmup 1 "${common[@]}"
      trace stream_n400000 python3 bench.py --samples 400000 --storage stream --steps 3 --warmup 1 "${common[@]}" ;;
    al)        # AdaGrad on the augmented Lagrangian at config 2's size and at the headline size
      trace adagrad_n20000 python3 bench.py --solver adagrad --samples 20000 --features 64 --steps 2000 --warmup 50 "${common[@]}"
      trace adagrad_n100000 python3 bench.py --solver adagrad --steps 30 --warmup 5 "${common[@]}" ;;
    as_c2)     # dense ActiveSet to 'optimal' at config 2's shape (kept factor, sweeps, Schur slots)
      trace as_c2 python3 bench.py --solver as --samples 20000 --features 64 --no-cpu ;;
    smo)
      trace smo_n100000 python3 bench.py --solver smo --no-cpu ;;
    gram_counters)
      tools/profile_gram_counters.sh "$dir/gram_counters" > "$out/gram_counters_stage.txt" 2>&1 || tail -5 "$out/gram_counters_stage.txt"
      cp "$out/gram_counters/gram_counters.txt" "$out/gram_counters.txt" 2>/dev/null ;;
    *) echo "[profile_all] unknown stage $stage" ;;
  esac
done
python3 tools/kernel_table.py "$out" > "$out/kernel_table.csv" 2> "$out/kernel_table.err" || tail -5 "$out/kernel_table.err"
wc -l "$out/kernel_table.csv"
